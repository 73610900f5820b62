"""Soak: N training steps of the headline configuration (HIP graphs on), watching loss finiteness and allocator growth.
python tools/soak.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    model, opt = bench.build_model(1024, 4, "skitG")
    opt.use_hip_graph = True
    # alternating 8-bit batches in pinned memory: the fresh-batch path of a train.py loop (one-launch image preparation, staged uploads)
    batches = [bench.make_batch(1024, 4, k, opt.style_code_dim, quantize8=True) for k in (0, 1, 2)]
    batches = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
    mem = []
    for it in range(steps):
        model.set_input(batches[it % 3], phase="train")
        model.optimize_parameters(epoch=1)
        if it % 100 == 99 or it == 4:
            torch.cuda.synchronize()
            losses = model.get_current_losses()
            ok = all(v == v and abs(v) < 1e30 for v in losses.values())
            mem.append(torch.cuda.memory_reserved() >> 20)
            print("step %4d  finite %s  reserved %d MiB  G_GAN %.3f D_real %.3f G_L1 %.3f" % (
                it + 1, ok, mem[-1], losses["l_G_GAN"], losses["l_D_real_I"], losses["l_G_L1"]), flush=True)
            assert ok
    assert mem[-1] <= mem[1] * 1.05 + 64, "allocator keeps growing: %s" % mem
    print("soak ok")


if __name__ == "__main__":
    main()
