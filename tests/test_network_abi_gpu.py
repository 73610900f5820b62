"""The network-level C entry vts_unet_forward (include/vts.h; SURVEY 8b's `vts_unet_fwd`): the generator's inference forward as ONE call.

  * against the Python schedule of the product (vts/engine.py:unet_forward), bit for bit, at the sizes whose inner layers take the
    k-split path and at the headline size; with the tiled style code of the skitG generator
  * against the REFERENCE module's output (tests/golden/nets_256.npz: CustomUnetGenerator run on CPU), within the north_star tolerance
  * from a host that is not Python: examples/unet_infer_host.cpp (built by __graft_entry__.build) reads the weights and the input from
    a file, runs the forward with hipMalloc'ed buffers on its own stream and writes the output -- equal to the product's, bit for bit
"""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, nets  # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "visual-tactile-synthesis_amd", "bin", "unet_infer_host")
FLAGS = ("--model %s --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False "
         "--lambda_G2_GAN_feat 0 --checkpoints_dir /tmp/vts_test_ckpt --name t --crop_size %d --batch_size %d")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def generator(size, n, model_name="sinskitG", seed=77):
    from models import create_model
    from options.train_options import TrainOptions

    opt = TrainOptions(cmd_line=FLAGS % (model_name, size, n)).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    G = model.netG
    if model_name == "sinskitG":
        G.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    else:
        G.load_state_dict(detrand.test_weights(nets.g_param_shapes(style_nc=opt.style_code_dim, num_layer_style_code=opt.num_layer_style_code), seed))
    return G, opt


@pytest.mark.parametrize("size,n", [(256, 1), (512, 2), (1024, 4)])
def test_c_forward_equals_the_python_schedule_bit_for_bit(size, n):
    from vts import engine

    G, _ = generator(size, n)
    dev = torch.device("cuda:0")
    s = detrand.uniform((n, 1, size, size), 5, "sketch").to(dev)
    grid = detrand.uniform((n, 8, size, size), 5, "grid").to(dev)
    y_py, _ = engine.unet_forward(G, (s, grid), keep=False)
    y_c = engine.unet_forward_c(G, (s, grid))
    torch.cuda.synchronize()
    assert torch.equal(y_c, y_py)
    # one source instead of a concatenated pair
    x = torch.cat([s, grid], 1)
    assert torch.equal(engine.unet_forward_c(G, x), y_py)
    # the tactile branch on a second stream (forked from / joined into the launch stream by events)
    y_l = engine.unet_forward_c(G, (s, grid), side_stream=torch.cuda.Stream())
    torch.cuda.synchronize()
    assert torch.equal(y_l, y_py)


def test_c_forward_matches_the_reference_module(golden_dir):
    """tests/golden/nets_256.npz holds the output of the reference's CustomUnetGenerator (run on CPU) for seeded weights and input"""
    from vts import engine

    g = np.load(os.path.join(golden_dir, "nets_256.npz"))
    size, seed = int(g["size"]), int(g["seed"])
    G, _ = generator(size, 1, seed=seed)
    G.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    x = detrand.uniform((1, 9, size, size), seed, "g_in").to("cuda:0")
    y = engine.unet_forward_c(G, x)
    assert rel(y[:, :, ::4, ::4], g["G_out_sub"]) < 1e-4          # north_star: outputs within 1e-3 rel-L2; observed ~1e-6


def test_c_forward_with_the_tiled_style_code():
    """skitG: the unit-norm style code tiled over the innermost map and concatenated there (style_code_mode concat, mapping tile)"""
    from vts import engine

    G, opt = generator(256, 2, model_name="skitG")
    dev = torch.device("cuda:0")
    x = detrand.uniform((2, 9, 256, 256), 9, "g_in").to(dev)
    sc = detrand.uniform((2, opt.style_code_dim), 9, "style")
    sc = (sc / sc.norm(dim=1, keepdim=True)).to(dev)
    y_py, _ = engine.unet_forward(G, x, style_code=sc, keep=False)
    tile = sc[:, :, None, None].expand(-1, -1, 256 >> G.num_downs, 256 >> G.num_downs).contiguous()
    assert torch.equal(engine.unet_forward_c(G, x, style_tile=tile), y_py)


def test_inference_with_every_decoder_layer_separate_takes_the_python_schedule():
    """num_layer_separate == num_downs (the reference asserts 0 <= nls <= nd, models/networks.py:1430-1645, and builds up{nd-1}_T): the C
    entry refuses a generator without a shared decoder trunk, so the inference forward must fall back to the Python schedule, not raise"""
    from models.networks import CustomUnetGenerator
    from vts import engine

    dev = torch.device("cuda:0")
    G = CustomUnetGenerator(9, 5, num_downs=8, ngf=10, num_layer_separate=8).to(dev)
    sd = detrand.test_weights(nets.g_param_shapes(num_layer_separate=8), 91)
    G.load_state_dict(sd)
    G.eval()
    assert not engine.unet_c_ok(G, None)
    x = detrand.uniform((1, 9, 256, 256), 6, "x")
    y = engine.unet_forward_infer(G, x.to(dev))
    ref = nets.unet_forward(sd, x, num_layer_separate=8)
    ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    torch.cuda.synchronize()
    assert rel(y, ref) < 1e-3       # north_star tolerance (rel-L2, fp32)


def test_bad_descriptors_are_refused_through_the_abi():
    import ctypes as C

    from vts import engine, lib as L

    G, _ = generator(256, 1)
    x = torch.zeros(1, 9, 256, 256, device="cuda:0")
    out = torch.empty(1, 5, 256, 256, device="cuda:0")
    lib = L.load()
    d = engine.unet_desc(G, x, out)
    d.H = 250                                    # not divisible by 2^num_downs (the reference's U-Net fails in torch.cat there)
    assert lib.vts_unet_forward_ws_floats(C.byref(d)) == -1 and b"divisible" in lib.vts_last_error()
    d = engine.unet_desc(G, x, out)
    need = lib.vts_unet_forward_ws_floats(C.byref(d))
    ws = torch.empty(16, device="cuda:0")
    assert need > 16 and lib.vts_unet_forward(C.byref(d), ws.data_ptr(), 16, None) != 0 and b"workspace" in lib.vts_last_error()
    d.up_cout[3] = 7                             # decoder / encoder channel mismatch
    assert lib.vts_unet_forward_ws_floats(C.byref(d)) == -1 and b"up3" in lib.vts_last_error()


def write_host_input(path, G, s, grid):
    nd, nls = G.num_downs, G.num_layer_separate
    n, _, h, w = s.shape
    sd = {k: v.detach().float().cpu().numpy() for k, v in G.state_dict().items()}
    with open(path, "wb") as f:
        f.write(struct.pack("<9i", 0x55535456, n, h, w, nd, nls, s.shape[1], grid.shape[1], 0))
        chans = [sd["down%d.model.%d.weight" % (i, 0 if i == 0 else 1)].shape[0] for i in range(nd)]
        upc = [sd["up%d.model.1.weight" % i].shape[1] for i in range(nd)]
        uptc = [sd["up%d_T.model.1.weight" % i].shape[1] if i < nls else 0 for i in range(nd)]
        for lst in (chans, upc, uptc):
            f.write(struct.pack("<%di" % nd, *lst))
        f.write(s.cpu().numpy().astype("<f4").tobytes())
        f.write(grid.cpu().numpy().astype("<f4").tobytes())
        for i in range(nd):
            k = "down%d.model.%d." % (i, 0 if i == 0 else 1)
            names = [k + "weight", k + "bias", "up%d.model.1.weight" % i, "up%d.model.1.bias" % i]
            if i < nls:
                names += ["up%d_T.model.1.weight" % i, "up%d_T.model.1.bias" % i]
            for name in names:
                f.write(np.ascontiguousarray(sd[name]).astype("<f4").tobytes())


@pytest.mark.skipif(not os.path.exists(HOST), reason="examples/unet_infer_host.cpp not built (python -c 'import __graft_entry__ as g; g.build()')")
def test_a_cpp_host_runs_the_generator_without_python(tmp_path):
    from vts import engine

    size, n = 512, 2
    G, _ = generator(size, n)
    dev = torch.device("cuda:0")
    s = detrand.uniform((n, 1, size, size), 3, "sketch").to(dev)
    grid = detrand.uniform((n, 8, size, size), 3, "grid").to(dev)
    y_py, _ = engine.unet_forward(G, (s, grid), keep=False)
    torch.cuda.synchronize()
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_host_input(fin, G, s, grid)
    r = subprocess.run([HOST, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "ms per image" in r.stdout
    y = torch.from_numpy(np.fromfile(fout, dtype="<f4").reshape(n, 5, size, size))
    assert torch.equal(y, y_py.cpu())


# ---- vts_patchgan_forward / vts_msd_forward (round 6; SURVEY 8b's `vts_msd_fwd`): the discriminators' training-mode forward as one C call ----

def _discriminator(input_nc, n_layers=3, seed=31):
    from models import networks

    D = networks.MultiscaleDiscriminator(input_nc, ndf=8, n_layers=n_layers, num_D=3).to("cuda:0")
    sd = detrand.test_weights(nets.d_param_shapes(input_nc, n_layers=n_layers), seed)
    D.load_state_dict(sd)
    return D, sd


def _buffers(D):
    return {k: b.detach().clone() for k, b in D.named_buffers()}


@pytest.mark.parametrize("shape,n_layers", [((2, 256, 256), 3), ((4, 130, 98), 3), ((96, 32, 32), 3), ((2, 128, 128), 2), ((3, 64, 64), 4)])
def test_patchgan_c_forward_equals_the_python_schedule_bit_for_bit(shape, n_layers):
    """every scale of the multiscale discriminator: prediction map, BatchNorm running statistics and num_batches_tracked after the call,
    and the recorded batch statistics of a stat-only pass -- C entry against vts/engine.py:_msd_scale_forward (full-size maps: tiled
    kernels with epilogue statistics; 32 x 32 patch stacks: small-map kernels; concat of two sources)"""
    from vts import engine
    from vts.ops import Act

    n, h, w = shape
    dev = torch.device("cuda:0")
    x0 = detrand.uniform((n, 1, h, w), 41, "s").to(dev)
    x1 = detrand.uniform((n, 3, h, w), 41, "i").to(dev)
    for stat_only in (False, True):
        res = {}
        for which in ("py", "c"):
            D, _ = _discriminator(4, n_layers)
            pyr = engine._pyramid(D, x0, x1)
            preds, recs = [], []
            for s in range(D.num_D):
                rec = {} if stat_only else None
                if which == "py":
                    acts = engine._msd_scale_forward(D, s, pyr[s][0], pyr[s][1], not stat_only, None, None, rec, None)
                    preds.append(acts[-1].data)
                else:
                    preds.append(engine.patchgan_forward_c(D, s, pyr[s][0], pyr[s][1], not stat_only, rec))
                recs.append(rec)
            torch.cuda.synchronize()
            res[which] = (preds, _buffers(D), recs)
        # depth 4 has a 64 -> 128 layer: the Python schedule sends it to the GEMM-class kernel (engine._flat4), the C entry keeps the 4x4
        # family -- the product does not take the C entry there (patchgan_c_ok); the entry itself is still correct to rounding
        exact = all(engine.patchgan_c_ok(D, pyr[s][0].data.shape[2], pyr[s][0].data.shape[3]) for s in range(D.num_D))
        assert exact == (n_layers <= 3)
        same = torch.equal if exact else (lambda u, v: rel(u, v) < 2e-5)
        for a, b in zip(res["py"][0], res["c"][0]):
            assert a.shape == b.shape and same(a, b)
        for k, v in res["py"][1].items():
            assert same(v.float(), res["c"][1][k].float()), k
        if stat_only:
            for ra, rb in zip(res["py"][2], res["c"][2]):
                assert sorted(ra) == sorted(rb) and len(ra) == n_layers
                for ci in ra:
                    assert same(ra[ci][0], rb[ci][0]) and same(ra[ci][1], rb[ci][1])
            for k, v in res["c"][1].items():      # a stat-only pass leaves the running buffers alone
                if k.endswith("num_batches_tracked"):
                    assert int(v) == 0, k
    # without the head: the statistics advance, no prediction map
    D, _ = _discriminator(4, n_layers)
    pyr = engine._pyramid(D, x0, x1)
    assert engine.patchgan_forward_c(D, 0, pyr[0][0], pyr[0][1], True, None, run_head=False) is None
    assert int(dict(D.named_buffers())["layer%d.3.num_batches_tracked" % (D.num_D - 1)]) == 1


def test_msd_c_forward_matches_oracle_and_the_per_scale_entry():
    """vts_msd_forward (all scales + the average-pool pyramid in one call) == the per-scale entry on the engine's pyramid, bit for bit, and
    == the oracle's MultiscaleDiscriminator (pinned to the reference module: tests/golden/nets_256.npz) within the north_star tolerance,
    predictions and BatchNorm buffers"""
    import ctypes as C

    from vts import engine
    from vts import lib as L

    dev = torch.device("cuda:0")
    n, h, w = 2, 256, 256
    x = detrand.uniform((n, 7, h, w), 43, "stack")
    D, sd = _discriminator(7)
    pyr = engine._pyramid(D, x.to(dev), None)
    want = [engine.patchgan_forward_c(D, s, pyr[s][0], None) for s in range(D.num_D)]
    bufs_scale = _buffers(D)
    D2, _ = _discriminator(7)
    d = L.MsdDesc()
    d.num_D = D2.num_D
    preds = []
    xd = x.to(dev)
    for s in range(D2.num_D):
        sub, pred, _ = engine.patchgan_desc(D2, s, pyr[s][0], None)      # (shapes per scale; only scale 0's input is read by the call)
        if s == 0:
            sub.in0 = L.operand(xd)
        d.scale[s] = sub
        preds.append(pred)
    lib = L.load()
    need = int(lib.vts_msd_forward_ws_floats(C.byref(d)))
    assert need > 0
    ws = torch.empty(need, device=dev)
    L.check(lib.vts_msd_forward(C.byref(d), ws.data_ptr(), ws.numel(), L.stream()), "vts_msd_forward")
    torch.cuda.synchronize()
    for a, b in zip(preds, want):
        assert torch.equal(a, b)
    for k, v in _buffers(D2).items():
        assert torch.equal(v, bufs_scale[k]), k
    # the oracle (CPU, pinned to the reference): training-mode forward, running statistics advanced once
    ref = nets.msd_forward(sd, x, 3)
    for s in range(3):
        assert rel(preds[s], ref[s][-1]) < 1e-4, s
    for k, v in _buffers(D2).items():
        if v.dtype.is_floating_point:
            assert rel(v, sd[k]) < 1e-4, k
        else:
            assert int(v) == int(sd[k]), k
    # argument errors are reported, not crashed
    assert lib.vts_patchgan_forward(None, None, 0, None) == -1 and b"null descriptor" in lib.vts_last_error()
    bad = L.PatchganDesc()
    assert lib.vts_patchgan_forward(C.byref(bad), ws.data_ptr(), ws.numel(), L.stream()) == -1
    assert lib.vts_msd_forward(C.byref(d), ws.data_ptr(), 8, L.stream()) == -1 and b"workspace" in lib.vts_last_error()


def test_forward_only_discriminator_passes_of_the_step_take_the_c_entry(monkeypatch):
    """the product's forward-only passes (D2 visualisation pass, D2 term of the generator step) through vts_patchgan_forward give the
    step of the Python schedule bit for bit: weights, BatchNorm buffers, logged losses, the visualised prediction map"""
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from torch.utils.data import default_collate
    from vts import engine

    res = {}
    calls = {"n": 0}
    orig = engine.patchgan_forward_c

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(engine, "patchgan_forward_c", counted)
    for use_c in (True, False):
        monkeypatch.setattr(engine, "MSD_C", use_c)
        calls["n"] = 0
        opt = TrainOptions(cmd_line=FLAGS % ("sinskitG", 256, 2)).parse()
        m = create_model(opt)
        m.setup(opt)
        m.parallelize()
        m.train()
        for net, shapes, seed in ((m.netG, nets.g_param_shapes(), 61), (m.netD, nets.d_param_shapes(4), 62), (m.netD2, nets.d_param_shapes(7), 63)):
            net.load_state_dict(detrand.test_weights(shapes, seed))
        import random
        random.seed(3)
        torch.manual_seed(3)
        batch = default_collate([make_sample(256, 64, 64, 70 + i) for i in range(2)])
        for _ in range(2):
            m.set_input(batch, phase="train")
            m.optimize_parameters(epoch=1)
        torch.cuda.synchronize()
        assert (calls["n"] > 0) == use_c
        res[use_c] = dict(flat={nm: getattr(m, "flat" + nm).flat.clone() for nm in ("G", "D", "D2")}, bufs=_buffers(m.netD2),
                          losses=m.get_current_losses(), viz=m.pred_fake_T_full.clone())
    a, b = res[True], res[False]
    for nm in a["flat"]:
        assert torch.equal(a["flat"][nm], b["flat"][nm]), nm
    for k in a["bufs"]:
        assert torch.equal(a["bufs"][k], b["bufs"][k]), k
    assert a["losses"] == b["losses"] and torch.equal(a["viz"], b["viz"])


MSD_HOST = os.path.join(ROOT, "visual-tactile-synthesis_amd", "bin", "msd_forward_host")


@pytest.mark.skipif(not os.path.exists(MSD_HOST), reason="examples/msd_forward_host.cpp not built (python -c 'import __graft_entry__ as g; g.build()')")
def test_a_cpp_host_runs_the_discriminator_forward_without_python(tmp_path):
    """examples/msd_forward_host.cpp: weights and input from a file, vts_msd_forward on hipMalloc'ed buffers, predictions and updated running
    statistics back -- equal to the product's forward (per-scale entry on the engine's pyramid) bit for bit"""
    from vts import engine

    n, c, h, w = 3, 7, 96, 80
    D, sd = _discriminator(c)
    dev = torch.device("cuda:0")
    x = detrand.uniform((n, c, h, w), 47, "stack")
    conv_idx = list(D.CONV_IDX)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<7i", 0x4453544d, n, c, h, w, D.num_D, len(conv_idx)))
        couts = [int(sd["layer0.%d.weight" % ci].shape[0]) for ci in conv_idx]
        f.write(struct.pack("<%di" % len(conv_idx), *couts))
        f.write(struct.pack("<%di" % len(conv_idx), *[D.STRIDE[ci] for ci in conv_idx]))
        f.write(struct.pack("<%di" % len(conv_idx), *[int(ci in D.BN_IDX) for ci in conv_idx]))
        f.write(np.ascontiguousarray(x.numpy()).astype("<f4").tobytes())
        for s in range(D.num_D):
            pre = "layer%d." % (D.num_D - 1 - s)
            for ci in conv_idx:
                names = ["%d.weight" % ci, "%d.bias" % ci]
                if ci in D.BN_IDX:
                    b = D.BN_IDX[ci]
                    names += ["%d.weight" % b, "%d.bias" % b, "%d.running_mean" % b, "%d.running_var" % b]
                for nm in names:
                    f.write(np.ascontiguousarray(sd[pre + nm].numpy()).astype("<f4").tobytes())
    r = subprocess.run([MSD_HOST, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "num_batches_tracked 1" in r.stdout
    pyr = engine._pyramid(D, x.to(dev), None)
    want = [engine.patchgan_forward_c(D, s, pyr[s][0], None) for s in range(D.num_D)]
    torch.cuda.synchronize()
    got = np.fromfile(fout, dtype="<f4")
    o = 0
    for s in range(D.num_D):
        k = want[s].numel()
        assert torch.equal(torch.from_numpy(got[o:o + k]).view_as(want[s]), want[s].cpu()), s
        o += k
    bufs = dict(D.named_buffers())
    for s in range(D.num_D):
        pre = "layer%d." % (D.num_D - 1 - s)
        for ci in conv_idx:
            if ci in D.BN_IDX:
                for nm in ("running_mean", "running_var"):
                    t = bufs["%s%d.%s" % (pre, D.BN_IDX[ci], nm)]
                    assert torch.equal(torch.from_numpy(got[o:o + t.numel()]), t.cpu()), (s, ci, nm)
                    o += t.numel()
    assert o == got.size
