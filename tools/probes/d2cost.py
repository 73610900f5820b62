import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/visual-tactile-synthesis_amd')
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
import torch, contextlib, io
import bench
from models import create_model
from options.train_options import TrainOptions
def run(extra):
    flags = ("--model skitG --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False --checkpoints_dir /tmp/vts_b --name b --crop_size 1024 --batch_size 4 " + extra)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = TrainOptions(cmd_line=flags).parse(); opt.gpu_ids=[0]
        m = create_model(opt); m.setup(opt); m.parallelize(); m.train()
    b = bench.make_batch(1024, 4, 0, opt.style_code_dim)
    m.set_input(b, phase="train")
    for _ in range(6): m.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(30): m.optimize_parameters(epoch=1)
    torch.cuda.synchronize(); print("%-40s %.3f ms" % (extra or "default", (time.perf_counter()-t)/30*1e3), flush=True)
    del m; torch.cuda.empty_cache()
run(""); run("--lambda_G2_GAN 0"); run("--use_more_fakeT False"); run("")
