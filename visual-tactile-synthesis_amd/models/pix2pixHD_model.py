"""pix2pixHD baseline on the HIP path (reference: models/pix2pixHD_model.py).

The reference trains this baseline patch-wise (`return_patch=True`: 32x32 sketch / image / tactile patches,
batch 32) with the coarse `GlobalGenerator`, two multiscale PatchGAN discriminators (D: sketch ++ image,
D2: sketch ++ tactile, both with `getIntermFeat`) and the LSGAN objective:

    forward            pix2pixHD_model.py:587-619     G(S) -> fake_I = out[:, :3] * M, fake_T = out[:, -2:] * M_T
    backward_D         :621-644                        0.5 (D_fake + D_real) + 0.5 (D2_fake + D2_real), one backward
    backward_G         :646-700                        G_GAN_I + G_GAN_T (+ feature matching + VGG)
    optimize_parameters :702-722                       D and D2 step, then G step

The GAN feature-matching term of the reference compares every discriminator feature with ITSELF (.detach(),
:662-680), so its value and gradient are identically zero: it is reported as 0.  The VGG feature term (VGGLoss on torchvision's VGG19,
networks.py:2021-2067) runs on the GEMM-class kernels (vts/perceptual.py); its pretrained weights cannot exist offline:
`--vgg_weights` loads them, otherwise seeded stand-ins are used and `loss_vgg_pretrained` is False.
"""
import os

import torch

from vts.misc import str2bool
from vts import engine, ops
from vts.ops import Act
from vts.optim import FlatAdam, FlatParams

from . import networks
from .base_model import BaseModel
from .sinskitG_model import add_model_flags

B = str2bool

# (flag, type, default[, choices])  -- reference: pix2pixHD_model.py:45-200
MODEL_FLAGS = [
    ("lambda_L1", float, 100.0), ("lr_G2", float, 0.0005), ("sketch_nc", int, 1), ("image_nc", int, 3), ("touch_nc", int, 2),
    ("data_len", int, 200), ("center_w", int, 1280), ("center_h", int, 960), ("num_touch_patch_for_logging", int, 10),
    ("use_bg_mask", B, True), ("T_resolution_multiplier", int, 1), ("padded_size", int, 1800), ("sample_bbox_per_patch", int, 2),
    ("save_S_patch", B, False), ("save_T_concat_tensor", B, False), ("save_raw_arr_vis", B, False), ("scale_nz", float, 0.25),
    ("return_patch", B, True), ("label_nc", int, 0), ("data_type", int, 32, [8, 16, 32]), ("no_instance", B, True),
    ("instance_feat", B, False), ("label_feat", B, False), ("feat_num", int, 3), ("load_features", "flag", False),
    ("n_downsample_E", int, 4), ("nef", int, 16), ("n_clusters", int, 10), ("n_downsample_global", int, 4),
    ("n_blocks_global", int, 9), ("n_blocks_local", int, 3), ("n_local_enhancers", int, 1), ("niter_fix_global", int, 0),
    ("getIntermFeat_D", B, True), ("num_D_D1", int, 2), ("num_D_D2", int, 2), ("no_gan_loss", B, False),
    ("no_ganFeat_loss", B, False), ("no_vgg_loss", B, False), ("lambda_feat", float, 10.0), ("lambda_vgg", float, 10.0),
    ("niter_decay", int, 100), ("separate_val_set", B, False), ("fp16", "flag", False),
]

LOSS_SLOTS = ["G_GAN_I", "G_GAN_T", "G_GAN", "D_real", "D_fake", "D2_real", "D2_fake", "G_GAN_Feat", "G_GAN_Feat_I", "G_GAN_Feat_T",
              "G_VGG", "G_VGG_I", "G_VGG_T"]


class Pix2PixHDModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        table = [r for r in MODEL_FLAGS if r[1] != "flag"]
        add_model_flags(parser, table)
        for name, _, _ in [r for r in MODEL_FLAGS if r[1] == "flag"]:
            parser.add_argument("--" + name, action="store_true", default=False)
        parser.add_argument("--use_hip_graph", type=B, default=True)   # not a reference flag: replay captured HIP graphs
        # not a reference flag: torchvision vgg19 state dict for VGGLoss (the reference downloads it, models/networks.py:2040); without a
        # file the term runs on seeded stand-in weights and `loss_vgg_pretrained` says so
        parser.add_argument("--vgg_weights", type=str, default="")
        parser.set_defaults(norm="batch", netG="global", netD="multiscale", ngf=64, dataset_mode="aligned", dataset="patchskit",
                            crop_size=1536, normG="instance", normD="instance", pool_size=0, n_epochs=50, n_epcohs_decay=150,
                            gan_mode="lsgan")
        verbose_freq = 320
        if is_train:
            parser.set_defaults(return_patch=True, batch_size=32, display_freq=verbose_freq, print_freq=verbose_freq,
                                save_latest_freq=verbose_freq, validation_freq=verbose_freq, save_epoch_freq=50, display_id=0,
                                save_raw_arr_vis=False)
        else:
            parser.set_defaults(return_patch=False, batch_size=1, save_S_patch=True, save_raw_arr_vis=False, sample_bbox_per_patch=1,
                                data_len=1)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        if not self.gpu_ids or not torch.cuda.is_available():
            raise RuntimeError("Pix2PixHDModel runs on the MI355X HIP path only (no CPU fallback): pass --gpu_ids 0 on a GPU box")
        self._check_unbuilt(opt)
        self.test_edit_S = "edit" in opt.dataroot
        self.model_names = ["G", "D", "D2"] if self.isTrain else ["G"]
        self.visual_names = ["real_S", "M", "fake_I", "fake_gx", "fake_gy", "fake_N"]
        if not self.test_edit_S:
            self.visual_names.insert(2, "real_I")
        self.loss_names = []
        if self.isTrain:
            if not opt.no_gan_loss:
                self.loss_names += ["G_GAN_I", "G_GAN_T", "G_GAN", "D_real", "D_fake", "D2_real", "D2_fake"]
            if not opt.no_ganFeat_loss:
                self.loss_names += ["G_GAN_Feat", "G_GAN_Feat_I", "G_GAN_Feat_T"]
            if not opt.no_vgg_loss:
                self.loss_names += ["G_VGG", "G_VGG_I", "G_VGG_T"]
        self.criterionGAN = networks.GANLoss(opt.gan_mode)
        self.netG = networks.define_G(opt.sketch_nc, opt.image_nc + opt.touch_nc, opt.ngf, opt.netG, opt.norm, gpu_ids=self.gpu_ids,
                                      opt=opt)
        self.flatG = FlatParams(self.netG)
        # data parallel: the generator's gradient travels as up to 8 buckets cut at layer boundaries, each started as soon as the backward
        # has written it (_segments); VTS_G_BUCKETS=1: one bucket behind the whole backward; VTS_G_BUCKET_MIN_MB: smallest bucket worth a collective
        if not getattr(self.netG, "is_local_enhancer", False):     # (the local enhancer's backward is a tree, not a chain: one bucket)
            self.flatG.chunk(int(os.environ.get("VTS_G_BUCKETS", "8")), min_floats=int(float(os.environ.get("VTS_G_BUCKET_MIN_MB", "16")) * (1 << 18)))
        if self.isTrain:
            self.netD = networks.define_D(opt.image_nc + opt.sketch_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, num_D=opt.num_D_D1,
                                          gpu_ids=self.gpu_ids, opt=opt)
            self.netD2 = networks.define_D(opt.touch_nc + opt.sketch_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm,
                                           num_D=opt.num_D_D2, gpu_ids=self.gpu_ids, opt=opt)
            self.flatD, self.flatD2 = FlatParams(self.netD), FlatParams(self.netD2)
            betas = (opt.beta1, 0.999)
            self.old_lr = opt.lr
            self.optimizer_G = FlatAdam(self.flatG, opt.lr, betas, span=self._finetune_span(opt))
            self.optimizer_D = FlatAdam(self.flatD, opt.lr, betas)
            self.optimizer_D2 = FlatAdam(self.flatD2, opt.lr, betas)
            self.optimizers += [self.optimizer_G, self.optimizer_D, self.optimizer_D2]
        if self.isTrain:
            if opt.pool_size > 0 and len(self.gpu_ids) > 1:
                raise NotImplementedError("Fake Pool Not Implemented for MultiGPU")      # (pix2pixHD_model.py:332-333)
            from util.image_pool import ImagePool
            self.fake_pool = ImagePool(opt.pool_size)
        self.netVGG = None
        if self.isTrain and not opt.no_vgg_loss:      # criterionVGG = VGGLoss(gpu_ids) (pix2pixHD_model.py; networks.py:2021-2067): frozen
            from . import perceptual
            self.netVGG = perceptual.build_vgg19(opt, self.device)
            self.loss_vgg_pretrained = bool(self.netVGG.pretrained)
        self._loss_buf = ops.loss_slots(len(LOSS_SLOTS), self.device)     # int64 fixed point (order-independent accumulation)
        self._slot = {n: self._loss_buf[i:i + 1] for i, n in enumerate(LOSS_SLOTS)}
        self._bufs = {}
        self._graphs = None
        self._eager_steps_done = 0
        self.ddp = None

    def _finetune_span(self, opt):
        """--niter_fix_global > 0 (pix2pixHD_model.py:403-421): optimizer_G is built over the parameters whose name starts with
        "model<n_local_enhancers>" only -- the last local enhancer, the tail of the flat buffer -- until update_fixed_params()."""
        if opt.niter_fix_global <= 0:
            return None
        prefix = "model" + str(opt.n_local_enhancers)
        lo, o, seen = None, 0, False
        for k, p in self.netG.named_parameters():
            hit = k.startswith(prefix)
            if hit and lo is None:
                lo = o
            if seen and not hit:
                raise RuntimeError("niter_fix_global: the finetuned parameters are not one contiguous range of the flat buffer")
            seen = seen or hit
            o += p.numel()
        if lo is None:     # netG 'global' has no such layer: the reference's Adam raises on the empty list
            raise ValueError("optimizer got an empty parameter list (--niter_fix_global needs --netG local)")
        print("------------- Only training the local enhancer network (for %d epochs) ------------" % opt.niter_fix_global)
        return (lo, o)

    def update_fixed_params(self):
        """pix2pixHD_model.py:942-949 (train.py:209-211 calls it at epoch niter_fix_global): a NEW Adam over all of netG -- fresh moments,
        step 0, the initial rate until the next update_learning_rate"""
        self.optimizer_G = FlatAdam(self.flatG, self.opt.lr, (self.opt.beta1, 0.999))
        self.optimizers[0] = self.optimizer_G
        self._drop_graphs()
        if self.opt.verbose:
            print("------------ Now also finetuning global generator -----------")

    def update_learning_rate(self):
        """pix2pixHD_model.py:951-962: every call lowers the rate of the three optimisers by opt.lr / niter_decay (the reference overrides
        BaseModel's scheduler step with this, and has no floor)"""
        lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        for o in (self.optimizer_D, self.optimizer_D2, self.optimizer_G):
            for g in o.param_groups:
                g["lr"] = lr
        if self.opt.verbose:
            print("update learning rate: %f -> %f" % (self.old_lr, lr))
        self.old_lr = lr

    @staticmethod
    def _check_unbuilt(opt):
        bad = []
        if not opt.no_instance or opt.instance_feat or opt.label_feat or opt.label_nc != 0 or opt.load_features:
            # not reachable in the reference either: its forward() hands encode_input inst = image = feat = None (pix2pixHD_model.py:592-597),
            # so --no_instance False dies on `inst_map.data` (:545-546), --instance_feat / --label_feat on netE.forward(None, ..) (:601-603)
            # and --load_features on `feat_map.data` (:558-559); --label_nc > 0 one-hots the SKETCH values (:533-541)
            bad.append("instance / label feature inputs (netE, label_nc > 0): the reference's forward() passes inst = image = feat = None "
                       "(models/pix2pixHD_model.py:592-603), so these flags fail upstream as well and there is no behaviour to restate")
        if opt.netG not in ("global", "local"):
            bad.append("netG %s (built: global, local)" % opt.netG)
        if opt.fp16 or opt.T_resolution_multiplier != 1 or not opt.use_bg_mask:
            bad.append("fp16 / T_resolution_multiplier != 1 / use_bg_mask False")
        if opt.isTrain and opt.no_gan_loss:
            bad.append("no_gan_loss")
        if bad:
            raise NotImplementedError("pix2pixHD on the HIP path: " + "; ".join(bad))

    # ------------------------------------------------------------------ input
    def _drop_graphs(self):
        if self._graphs is not None:
            self._graphs = None
            ops.release_ws((id(self), "train"))

    def _load(self, name, host):
        t = torch.as_tensor(host)
        buf = self._bufs.get(name)
        if buf is None or tuple(buf.shape) != tuple(t.shape):
            buf = self._bufs[name] = torch.empty(tuple(t.shape), dtype=torch.float32, device=self.device)
            self._drop_graphs()
        buf.copy_(t.to(torch.float32), non_blocking=True)
        return buf

    def set_input(self, input, phase="train", timing=False, verbose=False):
        """pix2pixHD_model.py:431-509: mask multiply; tactile patches reshaped to [N, 2, h, w] and masked."""
        self.data_phase = phase
        sk, mk, ik = ("S_images", "M_images", "I_images") if self.opt.return_patch else ("S", "M", "I")
        S = self._load("S", input[sk])
        self.M = self._load("M", input[mk])
        self.name = input["name"]
        self.image_paths = input["S_paths"]
        self.augmentation_params = input.get("augmentation_params")
        self.real_S = ops.mask_mul(S, self.M, out=S)
        if not self.test_edit_S:
            I = self._load("I", input[ik])
            self.real_I = ops.mask_mul(I, self.M, out=I)
            t = torch.as_tensor(input["T_images"])
            h, w = t.shape[-2:]
            T = self._load("T", t.reshape(-1, 2, h, w))
            masks = self._load("I_masks", torch.as_tensor(input["I_masks"]).reshape(-1, 1, h, w))
            self.real_T = ops.mask_mul(T, masks, out=T)
            self.real_gx, self.real_gy = self.real_T[:, 0:1], self.real_T[:, 1:2]

    # ------------------------------------------------------------------ forward
    def forward(self, infer=False, keep=False):
        n, _, h, w = self.real_S.shape
        dev = self.device
        g_out, self._g_ctx = engine.resnet_forward(self.netG, Act(self.real_S), keep=keep)
        self.g_out = g_out
        self.fake_I = torch.empty(n, 3, h, w, device=dev)
        self.fake_T = torch.empty(n, 2, h, w, device=dev)
        self.fake_N = torch.empty(n, 3, h, w, device=dev)
        ops.g_post(g_out, self.M, self.opt.scale_nz, fake_I=self.fake_I, fake_T=self.fake_T, fake_N=self.fake_N)
        self.fake_gx, self.fake_gy = self.fake_T[:, 0:1], self.fake_T[:, 1:2]

    def test(self, timing=False):
        with torch.no_grad():
            self.forward(keep=False)

    # ------------------------------------------------------------------ training step
    def _seg_forward_d(self):
        """forward, then backward_D: both discriminators, all scales side by side (engine.msd_multi)"""
        slot = self._slot
        self._loss_buf.zero_()
        self.forward(keep=True)

        def pair(fake, real, s_fake, s_real):
            return [dict(in0=self.real_S, in1=fake, real=False, coeff=1.0, slot=slot[s_fake], grad_coeff=0.5),
                    dict(in0=self.real_S, in1=real, real=True, coeff=1.0, slot=slot[s_real], grad_coeff=0.5, accumulate=True)]

        jobs_D = pair(self.fake_I, self.real_I, "D_fake", "D_real")
        if self.fake_pool.pool_size > 0:
            # backward_D's fake pass of D sees fake_pool.query(cat(label, fake_I)) (pix2pixHD_model.py:582, 626): label and image pooled
            # as parallel stores under one plan (drawn in optimize_parameters, outside the captured graphs)
            jobs_D[0]["in0"] = self.fake_pool.apply("label", self.real_S)
            jobs_D[0]["in1"] = self.fake_pool.apply("image", self.fake_I)
        engine.msd_multi([(self.netD, jobs_D),
                          (self.netD2, pair(self.fake_T, self.real_T, "D2_fake", "D2_real"))], self.criterionGAN)

    def _seg_adam_d_g(self):
        slot, dev = self._slot, self.device
        n, _, h, w = self.real_S.shape
        self.optimizer_D.step(self._gscale)
        self.optimizer_D2.step(self._gscale)
        d_fake_I = torch.empty(n, 3, h, w, device=dev)
        d_fake_T = torch.empty(n, 2, h, w, device=dev)
        engine.msd_multi([(self.netD, [dict(in0=self.real_S, in1=self.fake_I, real=True, coeff=1.0, slot=slot["G_GAN_I"], grad_coeff=1.0,
                                            param_grads=False, input_grad=(d_fake_I, False))]),
                          (self.netD2, [dict(in0=self.real_S, in1=self.fake_T, real=True, coeff=1.0, slot=slot["G_GAN_T"], grad_coeff=1.0,
                                             param_grads=False, input_grad=(d_fake_T, False))])], self.criterionGAN)
        slot["G_GAN"].copy_(slot["G_GAN_I"] + slot["G_GAN_T"])
        if self.netVGG is not None:
            # VGG feature matching (pix2pixHD_model.py:680-693): the image, and gx / gy each tiled to three channels
            from vts import perceptual as P_
            lam = self.opt.lambda_vgg
            d_fake_I.add_(P_.vgg_feature_l1(self.netVGG, self.fake_I, self.real_I, lam, slot["G_VGG_I"]))
            for c in (0, 1):
                f3 = self.fake_T[:, c:c + 1].expand(-1, 3, -1, -1).contiguous()
                r3 = self.real_T[:, c:c + 1].expand(-1, 3, -1, -1).contiguous()
                g3 = P_.vgg_feature_l1(self.netVGG, f3, r3, lam, slot["G_VGG_T"])
                ops.lpips_input_bwd(g3, (1.0, 1.0, 1.0), d_fake_T[:, c:c + 1], 1, accumulate=True)     # adjoint of the tiling: sum of the three
            slot["G_VGG"].copy_(slot["G_VGG_I"] + slot["G_VGG_T"])
        d_raw = torch.empty(n, 5, h, w, device=dev)
        ops.g_out_grad(d_fake_I, d_fake_T, self.M, self.g_out, d_raw)
        if self._g_stages():
            # data parallel, chunked generator bucket: only the LAST stage of the backward here (the layers of the last bucket)
            b = engine.resnet_stage_bounds(self._g_ctx, self.flatG.cut_params)
            self._g_bounds = b
            self._g_flow = engine.resnet_backward_stage(self.netG, self._g_ctx, d_raw, b[-2], b[-1])
        else:
            engine.resnet_backward(self.netG, self._g_ctx, d_raw)

    def _seg_g_stage(self, j):
        """stage j of the generator's backward (the layers of gradient bucket G_j), j = K-2 .. 0"""
        b = self._g_bounds
        self._g_flow = engine.resnet_backward_stage(self.netG, self._g_ctx, self._g_flow, b[j], b[j + 1])

    def _seg_adam_g(self):
        self.optimizer_G.step(self._gscale)

    def _g_stages(self):
        """number of stages of the generator's backward = chunks of its gradient bucket when the step runs data parallel, else 0"""
        from vts import ddp as _ddp
        if self.ddp is None or not _ddp.active() or not getattr(self.flatG, "cuts", None):
            return 0
        return len(self.flatG.cuts) + 1

    def _segments(self):
        """(segment, buckets to wait for before it, buckets to start after it).  Data parallel: the generator's gradient (730 MB at the
        reference's ngf 64) travels as K buckets cut at layer boundaries; the backward runs as K stages from the output layer down, and
        the all-reduce of bucket j starts as soon as stage j has written it, under the stages below."""
        K = self._g_stages()
        if not K:
            return [(self._seg_forward_d, (), ("D", "D2")), (self._seg_adam_d_g, ("D", "D2"), ("G",)), (self._seg_adam_g, ("G",), ())]
        segs = [(self._seg_forward_d, (), ("D", "D2")), (self._seg_adam_d_g, ("D", "D2"), ("G_%d" % (K - 1),))]
        for j in range(K - 2, -1, -1):
            segs.append((lambda j=j: self._seg_g_stage(j), (), ("G_%d" % j,)))
        segs.append((self._seg_adam_g, tuple("G_%d" % j for j in range(K - 1, -1, -1)), ()))
        return segs

    def _comm(self, name, start):
        if self.ddp is None or name not in self.ddp.buckets:
            return
        b = self.ddp.buckets[name]
        b.start() if start else b.wait()

    def _capture_graphs(self):
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream()
        counts = [o.step_count for o in self.optimizers]
        graphs, nodes = [], []
        ops.freeze_ws((id(self), "train"))
        try:
            for seg, _, _ in self._segments():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):   # RCCL's watchdog thread queries events meanwhile
                    seg()
                    nodes.append(ops.capture_node_count())
                graphs.append(g)
        except Exception:
            ops.release_ws((id(self), "train"))
            raise
        for o, c in zip(self.optimizers, counts):
            o.step_count = c
        self._graphs = graphs
        self.graph_nodes = nodes

    def optimize_parameters(self, epoch=0, timing=False):
        self._gscale = self.ddp.grad_scale if self.ddp is not None else 1.0
        if self.fake_pool.pool_size > 0:
            if self.ddp is not None:
                from vts import ddp as _ddp
                if _ddp.active():
                    raise NotImplementedError("Fake Pool Not Implemented for MultiGPU")
            self.fake_pool.next_batch(self.real_S.shape[0], self.device)
        for o in self.optimizers:
            o.sync_lr()
        use_graph = bool(getattr(self.opt, "use_hip_graph", False))
        if use_graph and self._graphs is None and self._eager_steps_done >= 1:
            self._capture_graphs()
        replay = use_graph and self._graphs is not None
        for i, (seg, wait_for, start_after) in enumerate(self._segments()):
            for nme in wait_for:
                self._comm(nme, start=False)
            if replay:
                self._graphs[i].replay()
            else:
                seg()
            for nme in start_after:
                self._comm(nme, start=True)
        if replay:
            for o in self.optimizers:
                o.step_count += 1
        else:
            self._eager_steps_done += 1

    # ------------------------------------------------------------------ logging
    def get_current_losses(self):
        vals = ops.loss_values(self._loss_buf)
        for i, name in enumerate(LOSS_SLOTS):
            setattr(self, "loss_" + name, vals[i])
        return BaseModel.get_current_losses(self)

    def compute_visuals(self):
        pass
