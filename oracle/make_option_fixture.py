"""Dump the reference's argparse surface (flag names, types, defaults) for the hot-path models into
tests/golden/ref_option_defaults.json.  TEST INFRASTRUCTURE: imports /root/reference (this container only).

    python -m oracle.make_option_fixture
"""
import argparse
import json
import os

from oracle import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_option_defaults.json")


def dump(model, is_train):
    from options.test_options import TestOptions
    from options.train_options import TrainOptions
    import models

    o = (TrainOptions if is_train else TestOptions)()
    base = o.initialize(argparse.ArgumentParser())
    base_dests = {a.dest for a in base._actions}
    parser = o.initialize(argparse.ArgumentParser())
    parser = models.get_option_setter(model)(parser, is_train)
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        t = a.type.__name__ if a.type is not None else ("flag" if a.nargs == 0 else "str")
        d = parser.get_default(a.dest)
        if isinstance(d, float) and d == float("inf"):
            d = "inf"
        out[a.dest] = {"default": d, "type": t, "nargs": a.nargs, "const": a.const, "choices": list(a.choices) if a.choices else None,
                       "model": a.dest not in base_dests}
    return out


def main():
    ref_import.load()
    res = {}
    for model in ("sinskitG", "skitG", "pix2pixHD"):
        for phase in ("train", "test"):
            res["%s_%s" % (model, phase)] = dump(model, phase == "train")
    with open(OUT, "w") as f:
        json.dump(res, f, indent=0, sort_keys=False)
    print("wrote", OUT, {k: len(v) for k, v in res.items()})


if __name__ == "__main__":
    main()
