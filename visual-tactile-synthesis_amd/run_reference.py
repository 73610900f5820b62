"""Run the REFERENCE's own train.py / test.py on the MI355X packages, without editing a reference file:

    python /path/to/visual-tactile-synthesis_amd/run_reference.py /path/to/reference/train.py --model sinskitG --gpu_ids 0 ...

`python /path/to/reference/train.py` with PYTHONPATH pointing here does NOT work: Python puts the script's directory at sys.path[0],
ahead of PYTHONPATH, so the reference's own data / models / options / util packages would be imported.  This launcher puts this
directory first, keeps the reference directory off the front of the path, and runs the script as __main__ (runpy does not add a
plain script's directory to sys.path).  Everything the two scripts import -- options.{train,test}_options, data.create_dataset,
models.create_model, util.visualizer.{Visualizer, save_images}, util.myhtml, util.util -- resolves here."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv):
    if len(argv) < 2 or not os.path.isfile(argv[1]):
        raise SystemExit("usage: run_reference.py /path/to/reference/{train,test}.py [reference flags ...]")
    script = os.path.abspath(argv[1])
    ref_dir = os.path.dirname(script)
    sys.path[:] = [HERE] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, ref_dir)]
    sys.argv = [script] + list(argv[2:])
    # torch's OpenMP workers spin after every host-side parallel region and starve the HIP runtime's completion thread (stalls of
    # 70..170 ms per step, tools/probes/stall_bisect2.py); the choice must be made before the script imports torch
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv)
