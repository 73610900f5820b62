"""SinSKITGModel on the MI355X HIP path: sketch -> (RGB image, tactile gx/gy) conditional GAN.

Drop-in for /root/reference/models/sinskitG_model.py behind the BaseModel contract:
same flags and defaults (:44-376), same loss / visual name lists (:401-470), same training
step order (optimize_parameters :601-700: forward -> patches -> D1 update -> D2 update ->
G update), same loss assembly (compute_D1_loss :1346-1407, compute_D2_loss :1409-1617,
compute_G1_loss :1660-1726, compute_G2_loss :1728-1842) and the same quirks (the G2 GAN
term is logged but carries no gradient :1751; feature matching is dead :1685,1794; the
nonsaturating loss ignores label smoothing; identity-size bicubic resamples are elided,
SURVEY.md §8a "Numerical conventions").

What is different on this side of the boundary:
  * every tensor op of the step is a libvts_hip.so kernel scheduled by vts.engine (no autograd);
  * batch size N >= 1 (the reference hard-requires 1): patches are gathered per sample;
  * losses stay on the device and are read back only in get_current_losses();
  * third-party terms whose weights cannot exist offline (LPIPS-VGG, CLIP vision-aided D3)
    must be disabled by flag -- requesting them raises instead of silently dropping them.
"""
import os
from vts import tune
import random

import numpy as np
import torch

from vts.misc import str2bool
from vts import engine, ops
from vts.ops import Act
from vts.optim import FlatAdam, FlatParams

from . import networks
from .base_model import BaseModel

# 1 (default): on one GPU the generator's discriminator-free loss terms run as one more lane beside the discriminator updates; 0: serially
# behind them (A/B timing; results are identical: the lanes only read the forward's outputs and add into their own fixed-point loss slots)
G_PRE_LANE = tune.get("VTS_G_PRE_LANE", "1") != "0"
D2_TAIL_LANE = tune.get("VTS_D2_TAIL_LANE", "1") != "0"    # joined / data-parallel schedule: D2's generator-step forward as a lane under the backward
FUSE_MERGE = tune.get("VTS_FUSE_MERGE", "1") != "0"       # last level of the D1 input-gradient pyramid merge inside g_out_grad
D2_CHAIN = tune.get("VTS_D2_CHAIN", "lanes")      # "serial": the whole D2 chain as one lane (measurement: see _run_d_chains)
D1_REAL_EARLY = tune.get("VTS_D1_REAL_EARLY", "1") != "0"     # D1's pass on the real images beside the generator forward (see _seg_d_updates)

B = str2bool

# (flag, type, default[, choices])  -- reference: sinskitG_model.py:52-296
MODEL_FLAGS = [
    ("use_cGAN", B, True), ("lambda_G1_GAN", float, 1.0), ("lambda_G1_L1", float, 100.0), ("lambda_G1_lpips", float, 1.0),
    ("use_cGAN_G2", B, True), ("use_cGAN_G2_S", B, True), ("use_cGAN_G2_I", B, True),
    ("lambda_G2_GAN", float, 5.0), ("lambda_G2_L1", float, 10.0), ("lambda_G2_lpips", float, 10.0),
    ("lambda_G2_GAN_feat", float, 1), ("smooth_GAN_label", "nargs_bool", True),
    ("use_vision_aided_loss", B, True), ("vision_aided_warmup_epoch", int, 100),
    ("lr_G2", float, 0.0005), ("netD2", str, "basic"), ("n_layers_D2", int, 3), ("num_layer_separate", int, 4),
    ("num_D_D2", int, 3), ("num_D_D1", int, 3), ("model_phase", str, "train"),
    ("sketch_nc", int, 1), ("image_nc", int, 3), ("touch_nc", int, 2),
    ("use_positional_encoding", B, True), ("positional_encoding_mode", str, "spe", ["spe", "csg"]),
    ("positional_encoding_dim", int, 4), ("data_len", int, 200), ("batch_size_G2", int, 64),
    ("batch_size_G2_val", int, 128), ("center_w", int, 1280), ("center_h", int, 960),
    ("T_resolution_multiplier", int, 1), ("padded_size", int, 1800), ("num_touch_patch_for_logging", int, 10),
    ("use_bg_mask", B, True), ("use_more_fakeT", B, True), ("add_fake_T_sample_size", int, 32),
    ("sample_bbox_per_patch", int, 2), ("use_diffaug", B, True), ("diffaugment", str, "bs"),
    ("w_resampling", B, True), ("resampling_w_min", int, 1), ("resampling_w_max", int, 10),
    ("save_S_patch", B, False), ("save_T_concat_tensor", B, False), ("save_raw_arr_vis", B, False),
    ("scale_nz", float, 0.25),
]

LOSS_SLOTS = ["G_GAN", "D_real_I", "D_fake_I", "D_I_grad_penalty", "G_L1", "G_lpips", "G2_GAN", "D_real_T_concat",
              "D_fake_T_concat", "D_T_grad_penalty", "D_more_fake_T", "G2_L1", "G2_lpips", "G2_GAN_feat", "G1_GAN_feat",
              "G_D3", "D3_real_I", "D3_fake_I"]


def add_model_flags(parser, table):
    for row in table:
        name, typ, default = row[0], row[1], row[2]
        if typ == "nargs_bool":
            parser.add_argument("--" + name, type=B, nargs="?", const=True, default=default)
        elif len(row) > 3:
            parser.add_argument("--" + name, type=typ, default=default, choices=row[3])
        else:
            parser.add_argument("--" + name, type=typ, default=default)


class SinSKITGModel(BaseModel):
    MODEL_NAME = "sinskitG"
    DATASET_MODE = "singleskit"
    DATAROOT = "./datasets/singleskit_FlowerShorts_padded_1800_x1/"
    DATA_LEN = 200

    @classmethod
    def modify_commandline_options(cls, parser, is_train=True):
        add_model_flags(parser, MODEL_FLAGS)
        cls.add_extra_flags(parser)
        # not a reference flag: replay the training step from captured HIP graphs (after one eager step)
        parser.add_argument("--use_hip_graph", type=B, default=True)
        # not a reference flag: state dict of torchvision's Inception-v3 (or pytorch-fid's) for the I_SIFID / T_SIFID metrics; the
        # reference downloads it (models/inception.py:58), which cannot happen offline
        parser.add_argument("--inception_weights", type=str, default="")
        # not a reference flag: state-dict file(s) of the LPIPS-VGG16 network (torchvision vgg16 features + lpips v0.1 lin layers, comma
        # separated); the reference gets them from `lpips.LPIPS(net="vgg")`, which downloads (sinskitG_model.py:495).  Without a file the
        # perceptual terms run on seeded stand-in weights and `loss_lpips_pretrained` / `metric_lpips_pretrained` say so.
        parser.add_argument("--lpips_weights", type=str, default="")
        # ... and of the AlexNet variant the reference evaluates with in the test phase (lpips.LPIPS(net="alex"), sinskitG_model.py:501:
        # torchvision alexnet features + lpips v0.1 lin layers)
        parser.add_argument("--lpips_alex_weights", type=str, default="")
        parser.set_defaults(model=cls.MODEL_NAME, dataset_mode=cls.DATASET_MODE, netG="unet256_custom", netD="multiscale",
                            netD2="multiscale", gan_mode="nonsaturating", ngf=10, ndf=8, lr=0.001, beta1=0.0, beta2=0.99,
                            crop_size=1536, no_flip=True, dataroot=cls.DATAROOT, data_len=cls.DATA_LEN)
        if is_train:
            parser.set_defaults(preprocess="crop", batch_size=1, display_freq=100, print_freq=100, save_latest_freq=100,
                                validation_freq=100, save_epoch_freq=50, n_epochs=5, n_epochs_decay=400, num_threads=0,
                                batch_size_G2=64, val_for_each_epoch=True, model_phase="train", display_id=0,
                                save_raw_arr_vis=False)
        else:
            parser.set_defaults(preprocess="none", batch_size=1, num_test=1, data_len=1, epoch="latest",
                                num_touch_patch_for_logging=100, batch_size_G2=100, model_phase="eval", display_id=0,
                                save_S_patch=True, save_raw_arr_vis=False, sample_bbox_per_patch=1)
        return parser

    @staticmethod
    def add_extra_flags(parser):
        pass

    # ------------------------------------------------------------------ construction
    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        if not self.gpu_ids or not torch.cuda.is_available():
            raise RuntimeError("%s runs on the MI355X HIP path only (no CPU fallback): pass --gpu_ids 0 on a GPU box"
                               % type(self).__name__)
        self._check_unbuilt_terms(opt)
        self.test_edit_S = "edit" in opt.dataroot
        self.model_names = ["G"]
        if self.isTrain:
            if opt.lambda_G1_GAN > 0.0:
                self.model_names.append("D")
            if opt.lambda_G2_GAN > 0.0:
                self.model_names.append("D2")
        self.visual_names = ["real_S", "M", "fake_I", "fake_gx", "fake_gy", "fake_N"]
        if not self.test_edit_S:
            self.visual_names.insert(2, "real_I")
        if self.isTrain and opt.lambda_G1_GAN > 0:
            self.visual_names.append("pred_fake_I")
        if self.isTrain and opt.lambda_G2_GAN > 0:
            self.visual_names.append("pred_fake_T_full")
        if opt.use_diffaug and not self.test_edit_S:
            self.visual_names.extend(["aug_fake_I", "aug_real_I"])
        self.loss_names = []
        if getattr(opt, "train_for_each_epoch", False):
            if opt.lambda_G1_GAN > 0.0:
                self.loss_names.extend(["G_GAN", "D_real_I", "D_fake_I", "D_I_grad_penalty"])
                if getattr(opt, "use_vision_aided_loss", False):      # (sinskitG_model.py:434-435; 0.0 until the warm-up epoch, :1399-1402, 1721-1722)
                    self.loss_names.extend(["G_D3", "D3_real_I", "D3_fake_I"])
            if opt.lambda_G1_L1 > 0.0:
                self.loss_names.append("G_L1")
            if opt.lambda_G2_GAN > 0.0:
                self.loss_names.extend(["G2_GAN", "D_real_T_concat", "D_fake_T_concat", "D_T_grad_penalty"])
                if opt.use_more_fakeT:
                    self.loss_names.append("D_more_fake_T")
            if opt.lambda_G2_L1 > 0.0:
                self.loss_names.append("G2_L1")
            if opt.lambda_G1_lpips > 0.0:
                self.loss_names.insert(self.loss_names.index("G_L1") + 1 if "G_L1" in self.loss_names else len(self.loss_names), "G_lpips")
            if opt.lambda_G2_lpips > 0.0:
                self.loss_names.append("G2_lpips")
        self.loss_G_D3 = self.loss_D3_real_I = self.loss_D3_fake_I = 0.0     # what the reference logs before the warm-up epoch
        # evaluation metrics (SIFID / LPIPS / PSNR / SSIM ...) are SURVEY.md §8(f) row 2: not built
        self.metric_names = []

        if opt.smooth_GAN_label:
            self.criterionGAN = networks.GANLoss(opt.gan_mode, target_real_label=0.8, target_fake_label=0.0)
        else:
            self.criterionGAN = networks.GANLoss(opt.gan_mode)
        if opt.gan_mode == "wgangp":
            raise NotImplementedError("gan_mode wgangp needs a double-backward gradient penalty; not built")

        self.pe_channels = 2 * opt.positional_encoding_dim if opt.use_positional_encoding else 0
        if opt.use_positional_encoding and opt.positional_encoding_mode != "spe":
            raise NotImplementedError("positional_encoding_mode %s is not built" % opt.positional_encoding_mode)
        input_nc = opt.sketch_nc + self.pe_channels
        if opt.netG in ("stylegan2", "smallstylegan2"):
            # the reference's StyleGAN2Decoder ends in ConvLayer(.., 3, 1) (stylegan_networks.py:892): three output channels cannot
            # feed this model's 3 + 2 channel split (the reference fails the same way at sinskitG_model.py:1309-1319); the generator
            # itself is built and tested as a network (networks.define_G, engine.sg2g_forward / sg2g_backward)
            raise NotImplementedError("--netG %s emits 3 channels (reference stylegan_networks.py:892); %s needs image_nc + touch_nc = %d"
                                      % (opt.netG, self.MODEL_NAME, opt.image_nc + opt.touch_nc))
        self.netG = networks.define_G(input_nc, opt.image_nc + opt.touch_nc, opt.ngf, opt.netG, opt.normG, not opt.no_dropout,
                                      opt.init_type, opt.init_gain, opt.no_antialias, opt.no_antialias_up, self.gpu_ids, opt,
                                      num_layer_separate=opt.num_layer_separate)
        # decoder parameters first in the flat buffers: their gradients form the bucket that is all-reduced under the encoder's backward
        self.flatG = FlatParams(self.netG, first=(lambda k: k.startswith("up")) if isinstance(self.netG, networks.CustomUnetGenerator) else None)
        self.use_cGAN_G2_S = bool(opt.use_cGAN_G2_S)
        self.use_cGAN_G2_I = bool(opt.use_cGAN_G2_I)
        # Conditioning of the discriminators (sinskitG_model.py:525-559): D1 sees cat(S, I) or, with --use_cGAN False, the image alone; the D2
        # stacks are [T(2)] + [S(1)] (use_cGAN_G2_S) + [I(3), mask(1)] (use_cGAN_G2_I).  Channel offsets of the stacks:
        self.use_cGAN = bool(opt.use_cGAN)
        self._cS = 2 if self.use_cGAN_G2_S else None                     # sketch channel
        self._cI = (2 + (1 if self.use_cGAN_G2_S else 0)) if self.use_cGAN_G2_I else None      # image channels (3), then the mask
        self._c2 = 2 + (1 if self.use_cGAN_G2_S else 0) + (4 if self.use_cGAN_G2_I else 0)
        if self.isTrain:
            if not opt.use_cGAN_G2:
                # probed on the reference (round 5, CPU): SinSKITGModel.__init__ dies -- define_D(...) for netD2 is called without `opt`
                # (sinskitG_model.py:575) and MultiscaleDiscriminator reads opt.gan_mode (networks.py:1658)
                raise NotImplementedError("--use_cGAN_G2 False: the reference itself cannot construct this model (define_D without opt, "
                                          "networks.py:1658); use --use_cGAN_G2_S False --use_cGAN_G2_I False for D2 on the tactile patches alone")
            if not opt.use_bg_mask:
                # probed on the reference: optimize_parameters dies at sinskitG_model.py:638 (self.M is only set under use_bg_mask, :721)
                raise NotImplementedError("--use_bg_mask False: the reference's training step cannot run without the mask (sinskitG_model.py:638 "
                                          "reads self.M, which set_input creates only under use_bg_mask, :721)")
            if opt.T_resolution_multiplier != 1:
                # Probed on the reference itself (round 3, CPU, 256 x 256, --T_resolution_multiplier 2, netG unet256_custom and
                # resnet_9blocks): its own optimize_parameters fails with "Sizes of tensors must match except in dimension 1. Expected
                # size 32 but got size 64" -- get_patch_in_input cuts (32 * multiplier)-pixel fake patches (sinskitG_model.py:1277) that
                # compute_D2_loss concatenates with the 32-pixel sketch / image / real patches (:1477-1484), and no generator of this
                # model emits the larger tactile map (networks.py:1101 needs generate_T_imgs).  The anti-aliased bicubic resampler those
                # branches call exists here (ops.bicubic_aa / bicubic_aa_bwd, pinned to F.interpolate(antialias=True)); there is no
                # working upstream behaviour to wire it to.
                raise NotImplementedError("T_resolution_multiplier %d: the reference's sinskitG step cannot run with a multiplier other than 1 "
                                          "(patch sizes 32 vs %d collide in compute_D2_loss, sinskitG_model.py:1477-1484)"
                                          % (opt.T_resolution_multiplier, 32 * opt.T_resolution_multiplier))
            if "D" in self.model_names:
                self.netD = networks.define_D(opt.image_nc + (opt.sketch_nc if self.use_cGAN else 0), opt.ndf, opt.netD, opt.n_layers_D, opt.normD,
                                              opt.init_type, opt.init_gain, opt.no_antialias, num_D=opt.num_D_D1,
                                              gpu_ids=self.gpu_ids, opt=opt)
                self.flatD = FlatParams(self.netD)
            if "D2" in self.model_names:
                self.netD2 = networks.define_D(self._c2, opt.ndf, opt.netD2,
                                               opt.n_layers_D2, opt.normD, opt.init_type, opt.init_gain, opt.no_antialias,
                                               num_D=opt.num_D_D2, gpu_ids=self.gpu_ids, opt=opt)
                self.flatD2 = FlatParams(self.netD2)
            betas = (opt.beta1, opt.beta2)
            # the device step counters of the three optimisers live in one tensor: advanced by ONE launch at the start of a step
            self._step_counters = torch.zeros(3, dtype=torch.int32, device=self.device)
            self.optimizer_G = FlatAdam(self.flatG, opt.lr, betas, step_dev=self._step_counters[0:1])
            self.optimizers.append(self.optimizer_G)
            if "D" in self.model_names:
                self.optimizer_D = FlatAdam(self.flatD, opt.lr, betas, step_dev=self._step_counters[1:2])
                self.optimizers.append(self.optimizer_D)
            if "D2" in self.model_names:
                self.optimizer_D2 = FlatAdam(self.flatD2, opt.lr_G2, betas, step_dev=self._step_counters[2:3])
                self.optimizers.append(self.optimizer_D2)
        # LPIPS-VGG16 (criterionLPIPS_vgg, sinskitG_model.py:495): frozen, not a saved network
        self.netLPIPS = None
        if self.isTrain and (opt.lambda_G1_lpips > 0 or opt.lambda_G2_lpips > 0):
            self._lpips_net()
        self._loss_buf = ops.loss_slots(len(LOSS_SLOTS), self.device)     # int64 fixed point (order-independent accumulation)
        self._slot = {n: self._loss_buf[i:i + 1] for i, n in enumerate(LOSS_SLOTS)}
        self._spe_cache = {}
        self._bufs = {}         # persistent input buffers (stable addresses for captured HIP graphs)
        self._pins, self._pin_evt = {}, {}   # pinned staging buffers of pageable host inputs (see _load)
        self._graphs = None     # the captured segments of the step, or None
        self._chained = False   # the chained single-segment schedule is in use (set by _segments)
        self._d2_lane = None
        self._infer_graph, self._infer_eager_done = None, False   # captured inference forward (test())
        self._eager_steps_done = 0
        if tune.get("VTS_KO_LANES", None) and tune.get("VTS_KO_LANES_ACK", "") != "timing-only":
            raise RuntimeError("VTS_KO_LANES skips discriminator lanes (wrong losses and gradients): set VTS_KO_LANES_ACK=timing-only to run the timing experiment")
        self._draws = None      # tests / parity runs inject {"aug": [4,N], "more_idx": [N,K]}
        self.ddp = None
        self.style_code = None

    @staticmethod
    def _check_unbuilt_terms(opt, epoch=None):
        """--use_vision_aided_loss (default True): the reference constructs vision_aided_loss.Discriminator(cv_type="clip",
        loss_type="multilevel_sigmoid_s") (sinskitG_model.py:546-551) but calls it only from epoch vision_aided_warmup_epoch (100) on
        (:1393, 1719); before that its three loss entries are the constant 0.0 and the step is exactly the step without the flag.  That
        part IS the HIP path's behaviour: the flag is accepted at construction (epoch None), the entries report 0.0, and
        optimize_parameters raises when the epoch reaches the warm-up epoch -- CLIP ViT-B/32 and the package's head exist neither offline
        nor in /root/reference."""
        if not opt.isTrain:
            return
        active = getattr(opt, "use_vision_aided_loss", False) and opt.lambda_G1_GAN > 0.0
        if epoch is None:
            last = getattr(opt, "n_epochs", 0) + getattr(opt, "n_epochs_decay", 0)
            if active and last >= opt.vision_aided_warmup_epoch:
                print("WARNING: --use_vision_aided_loss True with %d planned epochs: this build trains like the reference up to epoch %d and "
                      "then STOPS (the CLIP vision-aided discriminator terms that start at --vision_aided_warmup_epoch %d are not built; "
                      "the 'latest' checkpoint is written before the run stops).  Pass --use_vision_aided_loss False to train all epochs."
                      % (last, opt.vision_aided_warmup_epoch - 1, opt.vision_aided_warmup_epoch), flush=True)
            return
        if active and epoch >= opt.vision_aided_warmup_epoch:
            raise NotImplementedError(
                "epoch %d >= --vision_aided_warmup_epoch %d: from here on the reference adds the CLIP vision-aided discriminator terms "
                "(models/sinskitG_model.py:1393-1398, 1719-1720; third-party package vision_aided_loss + CLIP ViT-B/32 weights, neither "
                "available offline).  They are not built on the HIP path: continue with --use_vision_aided_loss False (note that the "
                "reference never hands netD3's head to an optimizer nor saves it, so the term is a frozen random projection of frozen "
                "CLIP features)." % (epoch, opt.vision_aided_warmup_epoch))

    def _lpips_alex_net(self):
        """lpips.LPIPS(net="alex"), the reference's eval_LPIPS of the test phase (sinskitG_model.py:501)"""
        if getattr(self, "netLPIPS_alex", None) is None:
            from . import perceptual
            self.netLPIPS_alex = perceptual.build_lpips_alex(self.opt, self.device)
        return self.netLPIPS_alex

    def _lpips_net(self):
        if self.netLPIPS is None:
            from . import perceptual
            self.netLPIPS = perceptual.build_lpips(self.opt, self.device)
            self.loss_lpips_pretrained = self.metric_lpips_pretrained = bool(self.netLPIPS.pretrained)
            if not self.netLPIPS.pretrained and self.isTrain and (self.opt.lambda_G1_lpips > 0 or self.opt.lambda_G2_lpips > 0):
                import sys
                print("WARNING: the LPIPS loss terms (lambda_G1_lpips %g, lambda_G2_lpips %g) run on SEEDED STAND-IN VGG16 weights: no --lpips_weights "
                      "file was given and the pretrained ones cannot be downloaded here.  Training with them does not reproduce the reference; "
                      "pass --lpips_weights <torchvision vgg16 + lpips v0.1 state dicts> or set both lambdas to 0."
                      % (self.opt.lambda_G1_lpips, self.opt.lambda_G2_lpips), file=sys.stderr, flush=True)
        return self.netLPIPS

    # ------------------------------------------------------------------ input
    def _buf(self, name, shape, dtype=torch.float32):
        """Persistent device buffer: inputs keep their addresses from batch to batch (captured HIP
        graphs read them in place; a shape change re-allocates and invalidates the graphs)."""
        t = self._bufs.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
            self._drop_graphs()
        return t

    @staticmethod
    def _stage(pin, src):
        """Fill a pinned staging view from a host tensor with ONE thread (numpy; casts on the way).  torch's CPU copy_ fans out over
        every core (128 OpenMP workers on the MI355X hosts) and their post-region spinning starved the HIP runtime's completion
        thread: every ~3rd graph-replayed step then stalled 70..170 ms (tools/probes/stall_bisect2.py: torch copy_ stalls, the numpy
        copy and torch.set_num_threads(1) do not)."""
        np.copyto(pin.numpy(), src.numpy() if torch.is_tensor(src) else np.asarray(src), casting="unsafe")

    def _load(self, name, host, dtype=torch.float32, staged=False):
        """host array -> persistent device buffer, asynchronously.  A pageable source would make the copy synchronous AND wait for the
        work already queued on the stream (the previous training step): such sources are staged through a persistent pinned buffer
        (a DataLoader with pin_memory=True hands over pinned tensors already; the small patch bookkeeping arrays never are).

        staged=True (the full-size S / I / M images): the H2D copy runs on a COPY STREAM into one of two device staging buffers, so the
        upload of batch i+1 travels over PCIe while the captured graphs of step i execute; the launch stream only waits for the copy's
        event, and what it reads from then on is the staging buffer (the masking kernels of set_input write the persistent tensors the
        graphs read).  A staging buffer is reused two batches later, after the event recorded at the end of the set_input that consumed
        it (_stage_done)."""
        t = torch.as_tensor(host)
        if staged:
            par = self._stage_parity
            buf = self._bufs.get("%s_stage%d" % (name, par))
            if buf is None or tuple(buf.shape) != tuple(t.shape) or buf.dtype != dtype:     # staging buffers are not read by graphs: no _drop_graphs
                buf = self._bufs["%s_stage%d" % (name, par)] = torch.empty(tuple(t.shape), dtype=dtype, device=self.device)
                # the caching allocator may hand back a block whose earlier launch-stream users are still queued: the copy stream
                # writes it right away, so it waits once for the launch stream (first allocation / re-allocation only)
                self._copy_stream.wait_stream(torch.cuda.current_stream())
        else:
            buf = self._buf(name, t.shape, dtype)
        src = t
        if t.device.type == "cpu" and not (t.is_pinned() and t.dtype == dtype):
            pin = self._pins.get(name)
            if pin is None or pin.shape != t.shape or pin.dtype != dtype:
                pin = self._pins[name] = torch.empty(t.shape, dtype=dtype).pin_memory()
                self._pin_evt[name] = torch.cuda.Event()
            else:
                self._pin_evt[name].synchronize()     # the previous upload from this staging buffer has been read
            self._stage(pin, t)
            src = pin
        if not staged:
            buf.copy_(src, non_blocking=True)
            if src is not t:
                self._pin_evt[name].record()
            return buf
        cs = self._copy_stream
        if self._stage_done[par] is not None:
            cs.wait_event(self._stage_done[par])      # the kernels that read this staging buffer two batches ago
        with torch.cuda.stream(cs):
            buf.copy_(src, non_blocking=True)
            if src is not t:
                self._pin_evt[name].record(cs)
        evt = torch.cuda.Event()
        evt.record(cs)
        torch.cuda.current_stream().wait_event(evt)
        return buf

    def _spe(self, n, h, w):
        key = (n, h, w)
        if key not in self._spe_cache:
            buf = torch.empty(n, self.pe_channels, h, w, dtype=torch.float32, device=self.device)
            ops.spe_grid(buf, self.opt.positional_encoding_dim)
            self._spe_cache = {key: buf}
            self._drop_graphs()
        return self._spe_cache[key]

    @staticmethod
    def _patch_offsets(coords):
        """find_coords_for_patch (models/model_utils.py:23-69) on the host, per sample."""
        c = np.asarray(coords, dtype=np.float64)
        ox = np.round(c[..., 0] + c[..., -2] / c[..., -3])
        oy = np.round(c[..., 1] + c[..., -1] / c[..., -3])
        cs = np.round(c[..., -4] / c[..., -3])
        return ox.astype(np.float32).astype(np.int32), oy.astype(np.float32).astype(np.int32), cs.astype(np.int32)

    def _patch_set(self, tag, T_images, I_masks, T_coords):
        """Upload one patch set (tactile squares, contact masks, gather offsets, image indices).  All five arrays travel as ONE
        pinned block and ONE H2D copy; the device tensors are views into one persistent block, filled by _stage (one host thread: the
        many-threaded torch copy_ this used to be is what stalled every ~3rd graph-replayed step, tools/probes/stall_bisect2.py)."""
        T = torch.as_tensor(T_images)
        n, nt = T.shape[0], T.shape[1]
        ox, oy, cs = self._patch_offsets(torch.as_tensor(T_coords).numpy())
        if not (cs == 32).all():
            raise NotImplementedError("patch cutout != 32 px (resize_ratio != 1): the reference asserts it never happens for this dataset "
                                      "(data/singleskit_dataset.py:874); the resampling branch of get_patch_in_input (model_utils.py:263-300) is not wired")
        P = n * nt
        f32 = torch.float32      # (the casts happen in _stage; .to() on the host would be another many-threaded torch op)
        parts = [("raw", T.reshape(P, 2, 32, 32), f32), ("masks", torch.as_tensor(I_masks).reshape(P, 1, 32, 32), f32),
                 ("offx", torch.from_numpy(np.ascontiguousarray(ox.reshape(-1))), torch.int32), ("offy", torch.from_numpy(np.ascontiguousarray(oy.reshape(-1))), torch.int32),
                 ("img", torch.from_numpy(np.repeat(np.arange(n, dtype=np.int32), nt)), torch.int32)]
        words = sum((t.numel() + 63) // 64 * 64 for _, t, _ in parts)       # all fields are 4-byte types; 256-byte aligned slots
        dev = self._buf(tag + "_block", (words,), torch.int32)
        # the pinned block travels on the COPY stream into one of two device staging blocks (the launch stream would otherwise read it over
        # PCIe, ~ 57 us per patch set, between two steps); the launch stream then copies device -> device into the block the graphs read
        par = getattr(self, "_stage_parity", 0)
        stage = self._bufs.get("%s_stage%d" % (tag, par))
        cs = getattr(self, "_copy_stream", None) if tune.get("VTS_PATCH_COPY_STREAM", "1") != "0" else None
        if cs is not None and (stage is None or stage.numel() != words):
            stage = self._bufs["%s_stage%d" % (tag, par)] = torch.empty(words, dtype=torch.int32, device=self.device)
            cs.wait_stream(torch.cuda.current_stream())
        pin = self._pins.get(tag)
        if pin is None or pin.numel() != words:
            pin = self._pins[tag] = torch.empty(words, dtype=torch.int32).pin_memory()
            self._pin_evt[tag] = torch.cuda.Event()
        else:
            while not self._pin_evt[tag].query():     # the previous upload from this staging block has been read (spin: no sleeping wait)
                pass
        views, o = {}, 0
        for name, t, dt in parts:
            k = t.numel()
            self._stage(pin[o:o + k].view(dt).view(t.shape), t)
            views[name] = dev[o:o + k].view(dt).view(t.shape)
            o += (k + 63) // 64 * 64
        from vts import lib as L
        if cs is None:
            L.check(L.load().vts_copy_words(pin.data_ptr(), dev.data_ptr(), words, L.stream()), "vts_copy_words")   # kernel reads the pinned block
            self._pin_evt[tag].record()
        else:
            if self._stage_done[par] is not None:
                cs.wait_event(self._stage_done[par])      # the device -> device copy that read this staging block two batches ago
            with torch.cuda.stream(cs):
                if tune.get("VTS_PATCH_COPY_KERNEL", "0") == "1":
                    L.check(L.load().vts_copy_words(pin.data_ptr(), stage.data_ptr(), words, L.stream()), "vts_copy_words")
                else:   # a DMA copy: a kernel on a fifth stream waits for one of the four hardware queues the step's lanes occupy
                    stage.copy_(pin, non_blocking=True)
                self._pin_evt[tag].record(cs)
                evt = torch.cuda.Event()
                evt.record(cs)
            torch.cuda.current_stream().wait_event(evt)
            dev.copy_(stage, non_blocking=True)
        real_T = ops.mask_mul(views["raw"], views["masks"], out=self._buf(tag + "_real_T", views["raw"].shape))
        return dict(real_T=real_T, masks=views["masks"], NT=nt, offx=views["offx"], offy=views["offy"], img=views["img"],
                    coords=np.asarray(T_coords))

    def set_input(self, input, phase="train", timing=False, verbose=False):
        self.data_phase = phase
        self.name = input.get("name")
        self.image_paths = input.get("S_paths")
        self.augmentation_params = input.get("augmentation_params")
        if not hasattr(self, "_copy_stream"):
            self._copy_stream, self._stage_parity, self._stage_done = torch.cuda.Stream(), 0, [None, None]
        self._stage_parity ^= 1
        # 8-bit sources (optional keys S_u8 / I_u8 / M_u8 next to the float tensors: the dataset front-ends attach them where the float
        # tensor IS ToTensor [+ Normalize] of those bytes): a quarter of the PCIe traffic, expanded on the device bit for bit (vts_u8_expand)
        u8 = tune.get("VTS_U8_BATCH", "1") != "0"

        def image(key, normalize):
            if u8 and (key + "_u8") in input:
                raw = self._load("%s_%s_u8" % (phase, key), input[key + "_u8"], dtype=torch.uint8, staged=True)
                par = self._stage_parity
                f = self._bufs.get("%s_%s_f%d" % (phase, key, par))
                if f is None or tuple(f.shape) != tuple(raw.shape):
                    f = self._bufs["%s_%s_f%d" % (phase, key, par)] = torch.empty(tuple(raw.shape), dtype=torch.float32, device=self.device)
                return ops.u8_expand(raw, normalize, out=f)
            return self._load("%s_%s" % (phase, key), input[key], staged=True)

        # the usual training batch (8-bit S / I / M, background mask, [fake | real] pair buffers): ONE launch writes M, both copies of the
        # masked sketch and the masked real image from the three staged byte tensors (vts_input_images_u8; seven launches otherwise)
        fused = (u8 and tune.get("VTS_FUSED_INPUT", "1") != "0" and self.opt.use_bg_mask and self.isTrain and phase == "train"
                 and "I" in input and all((k + "_u8") in input for k in ("S", "I", "M")))
        if fused:
            rawS, rawI, rawM = (self._load("%s_%s_u8" % (phase, k), input[k + "_u8"], dtype=torch.uint8, staged=True) for k in ("S", "I", "M"))
            S = rawS
        else:
            S = image("S", True)
        n, _, h, w = S.shape
        # The D1 update runs the discriminator on [fake | real] in ONE batched launch per layer (engine.msd_multi, `groups`):
        # sketch and image live in persistent [2n, C, H, W] buffers -- rows [0, n) are the fake pass (S, fake_I written by the
        # forward), rows [n, 2n) the real pass (S again, real_I).
        self._pair = bool(self.isTrain and phase == "train" and "I" in input)
        S2 = self._buf(phase + "_S2", (2 * n if self._pair else n, 1, h, w))
        self.real_S = S2[:n]
        if fused:
            self.M = self._buf(phase + "_M", tuple(torch.as_tensor(input["M"]).shape))
            self.M_T = self.M
            I2 = self._buf(phase + "_I2", (2 * n, 3, h, w))
            ops.input_images_u8(rawS, rawI, rawM, self.M, S2[:n], S2[n:], I2[n:])
        elif self.opt.use_bg_mask:
            self.M = self._buf(phase + "_M", tuple(torch.as_tensor(input["M"]).shape))      # read by the captured graphs: persistent; filled from the staging copy
            self.M.copy_(image("M", False))
            ops.mask_mul(S, self.M, out=self.real_S)
            self.M_T = self.M  # nearest resize at multiplier 1 is the identity
        else:
            self.real_S.copy_(S)
        if self._pair and not fused:
            S2[n:].copy_(self.real_S)
        self._S2 = S2
        if "I" in input:
            if fused:
                self.real_I = I2[n:]
            else:
                I = image("I", True)
                I2 = self._buf(phase + "_I2", (2 * n if self._pair else n, 3, h, w))
                self.real_I = I2[n:] if self._pair else I2
                if self.opt.use_bg_mask:
                    ops.mask_mul(I, self.M, out=self.real_I)
                else:
                    self.real_I.copy_(I)
            self._I2 = I2
            self.full_T_coords = input.get("full_T_coords")
            # Input pyramid of the multiscale D1 (AvgPool2d(3, 2, 1) per level, networks.py:1670,1692): the sketch and real-image levels
            # depend on the batch only, so they are pooled here, once; the fake-image rows are pooled once per step (_d1_pyramid) and
            # serve both the D update and the generator's GAN term (the reference pools all of them again in each of its three D1 calls).
            self._S2_pyr = self._I2_pyr = None
            if self._pair and "D" in self.model_names and hasattr(self.netD, "num_D") and not getattr(self.netD, "is_stylegan2_d", False):
                self._S2_pyr, self._I2_pyr = [S2], [I2]
                for s in range(1, self.netD.num_D):
                    hs, ws = (self._S2_pyr[-1].shape[2] - 1) // 2 + 1, (self._S2_pyr[-1].shape[3] - 1) // 2 + 1
                    Sp = ops.avgpool(self._S2_pyr[-1], y=self._buf("%s_S2_p%d" % (phase, s), (2 * n, 1, hs, ws)))
                    Ip = self._buf("%s_I2_p%d" % (phase, s), (2 * n, 3, hs, ws))
                    ops.avgpool(self._I2_pyr[-1][n:], y=Ip[n:])
                    self._S2_pyr.append(Sp)
                    self._I2_pyr.append(Ip)
        elif hasattr(self, "real_I"):
            del self.real_I
        self.S_pe = self._spe(n, h, w) if self.pe_channels else None
        self._style_tiles = None
        if "style_code" in input:
            self.style_code = self._load(phase + "_style", input["style_code"])
            G = self.netG
            if getattr(G, "use_style", False) and getattr(G, "style_mapping", "") == "tile":
                # the tiled style code (networks.py:1600-1623) depends on the batch only: tiled here, once per batch, into persistent buffers
                self._style_tiles = {}
                for i in range(G.num_downs - G.num_layer_style_code, G.num_downs):
                    hh, ww = h >> (i + 1), w >> (i + 1)
                    t = self._buf("%s_style_tile%d" % (phase, i), (n, self.style_code.shape[1], hh, ww))
                    t.copy_(self.style_code.to(torch.float32)[:, :, None, None].expand(-1, -1, hh, ww))
                    self._style_tiles[i] = t
        self.train_set = self.val_set = None
        if "T_images" in input and len(input["T_images"]) > 0:
            self.train_set = self._patch_set(phase + "_tr", input["T_images"], input["I_masks"], input["T_coords"])
            self.train_T_coords = self.train_set["coords"]
            self.train_real_T_concat = self.train_set["real_T"]
            self.train_I_masks = self.train_set["masks"]
            if "val_T_images" in input and len(input["val_T_images"]) > 0:
                # only compute_metrics reads the validation patches: in the training phase they are uploaded on first use (the `val_set`
                # property) instead of with every batch -- a host staging copy, a DMA and two launches per step saved
                if phase == "train" and tune.get("VTS_LAZY_VAL_SET", "1") != "0":
                    self._val_pending = (phase + "_va", input["val_T_images"], input["val_I_masks"], input["val_T_coords"])
                else:
                    self.val_set = self._patch_set(phase + "_va", input["val_T_images"], input["val_I_masks"], input["val_T_coords"])
            elif phase == "test":
                self.val_set = self.train_set
        if self.isTrain and self.opt.use_more_fakeT and phase == "train":
            # candidate positions of the "more fake T" sampler depend on the mask only: build them here,
            # where the host already synchronises for the H2D copies (model_utils.py:212-216)
            prev = getattr(self, "_cand_check", None)
            if prev is not None:
                torch.cuda.current_stream().wait_event(prev[1])     # (the previous batch's counts have left the buffer this call rewrites)
            self._cand, self._cand_prefix = ops.mask_candidates(
                self.M, self._buf("cand", (n, h - 14, w - 14), torch.uint8), self._buf("cand_prefix", (n, h - 14 + 1), torch.int32))
            k = self.opt.add_fake_T_sample_size
            # The reference draws random.sample(range(count), k) (model_utils.py:217) and RAISES when an image has fewer than k candidate
            # positions; the device-side draw wraps instead.  Reading the counts here would stall the host behind the step in flight, so
            # they travel to pinned memory asynchronously and are checked one set_input later (or at the next loss read), by which time
            # the copy has long finished: a too-small mask fails loudly, one step late, at no cost.
            self._check_candidate_counts()
            pin = self._bufs.get("cand_count_pin")
            if pin is None or pin.numel() != n:
                pin = self._bufs["cand_count_pin"] = torch.empty(n, dtype=torch.int32).pin_memory()
            if getattr(self, "_copy_stream", None) is not None and tune.get("VTS_CAND_COPY_STREAM", "1") != "0":
                # off the launch stream: a device -> host copy between set_input's kernels and the step's graphs costs the launch stream two
                # engine switches (~ 0.1 ms of idle device); nothing on the launch stream reads it
                ready = torch.cuda.Event()
                ready.record()
                self._copy_stream.wait_event(ready)
                with torch.cuda.stream(self._copy_stream):
                    pin.copy_(self._cand_prefix[:, -1], non_blocking=True)
                    evt = torch.cuda.Event()
                    evt.record(self._copy_stream)
            else:
                pin.copy_(self._cand_prefix[:, -1], non_blocking=True)
                evt = torch.cuda.Event()
                evt.record()
            self._cand_check = (pin, evt, k, self.name)
            self._ranks = self._buf("more_ranks", (n, k), torch.int64)
            mi = self._bufs.get("more_img")          # image index of every extra patch: a constant of (n, k), built on the device once
            if mi is None or mi.numel() != n * k:
                mi = self._buf("more_img", (n * k,), torch.int32)
                mi.copy_(torch.arange(n, dtype=torch.int32, device=self.device).repeat_interleave(k))
            self._more_img = mi
        done = torch.cuda.Event()
        done.record()        # every reader of this batch's staging buffers has been queued on the launch stream
        self._stage_done[self._stage_parity] = done

    @property
    def val_set(self):
        pend = getattr(self, "_val_pending", None)
        if pend is not None:
            self._val_pending = None
            self._val_set = self._patch_set(*pend)
        return getattr(self, "_val_set", None)

    @val_set.setter
    def val_set(self, v):
        self._val_pending = None
        self._val_set = v

    def _check_candidate_counts(self, wait=False):
        """raises like random.sample would have (reference models/model_utils.py:217) when the previous batch's mask had fewer candidate
        positions than add_fake_T_sample_size; never blocks unless `wait`"""
        chk = getattr(self, "_cand_check", None)
        if chk is None:
            return
        pin, evt, k, name = chk
        if not wait and not evt.query():
            return
        evt.synchronize()
        self._cand_check = None
        counts = pin.tolist()
        if min(counts) < k:
            raise ValueError("Sample larger than population: the background mask of %s leaves %s candidate positions for the %d 'more fake T' "
                             "patches (reference: random.sample in get_patch_in_input, models/model_utils.py:217)" % (name, counts, k))

    # ------------------------------------------------------------------ forward
    def _g_input(self):
        return (Act(self.real_S), Act(self.S_pe)) if self.S_pe is not None else Act(self.real_S)

    def forward(self, timing=False, keep=False):
        opt = self.opt
        n, _, h, w = self.real_S.shape
        dev = self.device
        if isinstance(self.netG, networks.ResnetGenerator):
            if self._style() is not None:
                raise NotImplementedError("style codes are only built for netG=unet256_custom")
            g_out, self._g_ctx = engine.resnet_forward(self.netG, self._g_input(), keep=keep,
                                                       dropout_masks=self._draws.get("dropout") if self._draws is not None else None)
        elif not keep:
            # inference: one call of the network-level C entry (include/vts.h: vts_unet_forward) where it covers the configuration
            g_out, self._g_ctx = engine.unet_forward_infer(self.netG, self._g_input(), style_code=self._style(),
                                                           style_tiles=getattr(self, "_style_tiles", None)), None
        else:
            g_out, self._g_ctx = engine.unet_forward(self.netG, self._g_input(), style_code=self._style(), keep=keep,
                                                     style_tiles=getattr(self, "_style_tiles", None),
                                                     dropout_masks=self._draws.get("dropout") if self._draws is not None else None)
        self.g_out = g_out
        has_real = hasattr(self, "real_I") and not self.test_edit_S
        self.fake_I = self._I2[:n] if (has_real and getattr(self, "_pair", False)) else torch.empty(n, 3, h, w, device=dev)
        self.fake_N = torch.empty(n, 3, h, w, device=dev)
        # the D2 full-resolution stack [fake_T(2), S(1), aug_fake_I(3), M(1)] (default conditioning; see _cS / _cI) is filled in place
        cS, cI = self._cS, self._cI
        self._full_stack = torch.empty(n, self._c2, h, w, device=dev)
        self.fake_T = self._full_stack[:, 0:2]
        aug_fake = (self._full_stack[:, cI:cI + 3] if cI is not None else torch.empty(n, 3, h, w, device=dev)) if has_real else None
        rb = rs = None
        policy = None
        if has_real and opt.use_diffaug and opt.diffaugment and opt.diffaugment != "bs":
            # any other policy over the reference's letters b s c t o n (thirdparty/DiffAugment.py:89-96): a chain of single-operation
            # launches instead of the fused 'bs' pass; the generator's post-processing then leaves aug_fake_I to that chain
            policy = opt.diffaugment
            if self._draws is not None and "aug_policy" in self._draws:
                pdraws = self._draws["aug_policy"]
            else:       # the reference's order: DiffAugment(real_I) draws first, then DiffAugment(fake_I) (:1330-1333)
                pdraws = (ops.diffaug_draws(policy, (n, 3, h, w), dev), ops.diffaug_draws(policy, (n, 3, h, w), dev))
            self.aug_real_I = ops.diffaug_policy(self.real_I, policy, pdraws[0], self.M, torch.empty_like(self.real_I))
        if policy is not None:
            pass
        elif has_real and opt.use_diffaug and opt.diffaugment:
            # (round 6, measured and dropped: the draws + this launch on the real image in a lane BESIDE the generator forward -- they need
            #  nothing of the generator -- instead of on the serial stretch behind it: 5.31 vs 5.28 ms, the lane competes with D1's early pass)
            draws = self._draws["aug"].to(dev).float() if self._draws is not None else torch.rand(4, n, device=dev)
            self._aug = draws
            rb, rs = draws[2].contiguous(), draws[3].contiguous()
            self.aug_real_I = ops.diffaug_bs_mask(self.real_I, self.M, draws[0].contiguous(), draws[1].contiguous(),
                                                  torch.empty_like(self.real_I))
        elif has_real:
            # no augmentation: aug_* are the masked images themselves (M is binary, so *M again is idempotent)
            half, one = torch.full((n,), 0.5, device=dev), torch.full((n,), 0.5, device=dev)
            rb, rs = half, one  # brightness shift 0, saturation factor 1
            self.aug_real_I = self.real_I
        # (training: the same pass writes the sketch and the mask into their channels of the full-resolution D2 stack)
        ops.g_post(g_out, self.M, opt.scale_nz, rb, rs, fake_I=self.fake_I, fake_T=self.fake_T, fake_N=self.fake_N,
                   aug_fake_I=aug_fake if policy is None else None, S=self.real_S if keep else None,
                   stack_S=self._full_stack[:, cS:cS + 1] if (keep and cS is not None) else None,
                   stack_M=self._full_stack[:, cI + 3:cI + 4] if (keep and cI is not None) else None)
        if policy is not None:
            ops.diffaug_policy(self.fake_I.contiguous(), policy, pdraws[1], self.M, aug_fake)
        self.aug_fake_I = aug_fake
        self.fake_gx = self.fake_T[:, 0:1]
        self.fake_gy = self.fake_T[:, 1:2]

    def _style(self):
        return None

    def test(self, timing=False):
        """inference forward; with --use_hip_graph the second call on unchanged shapes captures it as a HIP graph and
        later calls replay it (the ~45 launches of one image are otherwise host-latency bound)"""
        with torch.no_grad():
            if not getattr(self.opt, "use_hip_graph", False) or self._draws is not None:
                return self.forward(keep=False)
            if self._infer_graph is not None:
                self._infer_graph.replay()
                # a training replay in between rebinds fake_I / fake_T / ... to the TRAINING graphs' tensors (_graph_attrs):
                # re-attach the tensors this graph writes, or metrics / visuals of a validation pass would read the last training batch
                self.__dict__.update(self._infer_attrs)
                return
            if self._infer_eager_done:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                ops.freeze_ws((id(self), "infer"))
                try:
                    with torch.cuda.graph(g, stream=torch.cuda.Stream(), capture_error_mode="thread_local"):
                        self.forward(keep=False)
                except Exception:
                    ops.release_ws((id(self), "infer"))
                    raise
                self._infer_graph = g
                self._infer_attrs = {k: getattr(self, k) for k in self._INFER_OUTPUTS if hasattr(self, k)}
                g.replay()
                return
            self.forward(keep=False)
            self._infer_eager_done = True

    # ------------------------------------------------------------------ training step
    def _gather(self, src, pset, out, c0, channels=None):
        return ops.patch_gather(src, pset["img"], pset["offx"], pset["offy"], 32, out, c0=c0, channels=channels)

    def _prepare_ranks(self):
        """Host side of the 'more fake T' sampler (random.sample over the candidate list,
        model_utils.py:217): draws the ranks and uploads them into the persistent buffer."""
        if not (self.opt.use_more_fakeT and "D2" in self.model_names):
            return
        k = self.opt.add_fake_T_sample_size
        if self._draws is None and tune.get("VTS_HOST_RANKS", "0") != "1":
            # drawn on the device (vts_mask_sample_ranks: Floyd's algorithm over the candidate count the device already holds, seeded from
            # Python's `random` so that random.seed() still fixes the run).  The host used to fetch that count first -- a second evaluation
            # of the candidate map on the copy stream, a pinned read-back and a spin on its event, ~5 ms per iteration with a fresh batch
            # (tools/prof_fresh.py) -- and to upload the ranks it drew.
            ops.mask_sample_ranks(self._cand_prefix, self.M.shape[2], k, random.getrandbits(64), self._ranks)
            return
        if self._draws is not None:
            ranks = torch.as_tensor(self._draws["more_idx"]).long()
        else:
            counts = self._cand_prefix[:, -1].tolist()       # (VTS_HOST_RANKS=1: the host-side draw of round 2, synchronising)
            ranks = torch.tensor([random.sample(range(c), k) for c in counts], dtype=torch.int64)
        pin = self._bufs.get("ranks_pin")
        if pin is None or pin.shape != ranks.shape:
            pin = self._bufs["ranks_pin"] = torch.empty(ranks.shape, dtype=torch.int64).pin_memory()
            self._ranks_evt = torch.cuda.Event()
        else:
            self._ranks_evt.synchronize()
        self._stage(pin, ranks)
        self._ranks.copy_(pin, non_blocking=True)     # pinned source: a pageable one would make this copy wait for the queued work
        self._ranks_evt.record()

    def _d_pass(self, net, in0, in1, target_real, coeff, slot, accumulate, backward=True):
        """One discriminator forward (+ backward into its parameter grads).  Returns preds."""
        preds, ctx = engine.msd_forward(net, in0, in1, keep=backward)
        dp = self.criterionGAN.accumulate(preds, target_real, coeff, slot, grad_coeff=0.5 * coeff, want_grad=backward,
                                          pre_sigmoid=getattr(net, "use_sigmoid", False))
        if backward:
            engine.msd_backward(net, ctx, dp, param_grads=True, accumulate=accumulate)
        return preds

    # The step is cut into segments at the points where a data-parallel run exchanges gradients.
    # Each segment is pure device work on persistent buffers, so it can run eagerly or be replayed
    # from a captured HIP graph (optimize_parameters below).
    def _forward_and_stacks(self, begin=True):
        opt, dev, ts, slot = self.opt, self.device, self.train_set, self._slot
        P = ts["real_T"].shape[0]
        if begin:
            ops.step_begin(self._loss_buf, self._step_counters)      # loss slots <- 0, optimiser step counters += 1
        self.forward(keep=True)
        # patches (compute_additional_output :1268-1291)
        # the patch stacks of the D2 update in ONE buffer, [fake | more fake | real] along the batch (batched passes)
        K = self.real_S.shape[0] * opt.add_fake_T_sample_size if (opt.use_more_fakeT and "D2" in self.model_names) else 0
        cS, cI = self._cS, self._cI
        self._stack_all = torch.empty(2 * P + K, self._c2, 32, 32, device=dev)
        fake_stack = self._stack_all[:P]                      # [fake_T, S, aug_fake_I, mask]
        real_stack = self._stack_all[P + K:]                  # [real_T, S, aug_real_I, mask]
        self._more_stack = self._stack_all[P:P + K]
        self.fake_T_concat = torch.empty(P, 2, 32, 32, device=dev)
        g = dict(img=ts["img"], offx=ts["offx"], offy=ts["offy"])
        # every channel run of the three stacks in ONE launch (ops.patch_jobs; it was nine gathers and five copies)
        jobs = [dict(dst=fake_stack, c0=0, src=self.fake_T, channels=2, **g), dict(dst=real_stack, c0=0, src=ts["real_T"]),
                dict(dst=self.fake_T_concat, c0=0, src=self.fake_T, channels=2, **g)]
        if cS is not None:
            jobs += [dict(dst=fake_stack, c0=cS, src=self.real_S, **g), dict(dst=real_stack, c0=cS, src=self.real_S, **g)]
        if cI is not None:
            jobs += [dict(dst=fake_stack, c0=cI, src=self.aug_fake_I, channels=3, **g), dict(dst=fake_stack, c0=cI + 3, src=ts["masks"]),
                     dict(dst=real_stack, c0=cI, src=self.aug_real_I, **g), dict(dst=real_stack, c0=cI + 3, src=ts["masks"])]
        if K:
            # the "more fake T" squares at random positions of the dilated mask (model_utils.py:212-222): [fake_T, S, fake_I, 1]
            h, w = self.real_S.shape[2:]
            mox, moy = ops.mask_select(self._cand, self._cand_prefix, self._ranks, h, w)
            self.fake_sample_offset_x, self.fake_sample_offset_y = mox, moy
            m = dict(img=self._more_img, offx=mox, offy=moy)
            jobs += [dict(dst=self._more_stack, c0=0, src=self.fake_T, channels=2, **m)]
            if cS is not None:
                jobs += [dict(dst=self._more_stack, c0=cS, src=self.real_S, **m)]
            if cI is not None:
                jobs += [dict(dst=self._more_stack, c0=cI, src=self.fake_I, **m), dict(dst=self._more_stack, c0=cI + 3, channels=1, fill=1.0)]
        ops.patch_jobs(jobs)
        self._fake_stack, self._real_stack = fake_stack, real_stack

    def _seg_d_updates(self):
        """forward, then the D1 (full resolution) and D2 (32x32 patches) updates: all scales of both discriminators run
        side by side (engine.msd_multi); within one discriminator the passes keep the reference's order."""
        opt, dev, slot = self.opt, self.device, self._slot
        n = self.real_S.shape[0]
        # D1 on the REAL images does not depend on the generator: its three scales (forward, backward, weight gradients) run on side streams
        # BESIDE the generator forward, which is one chain that leaves most of the chip idle.  The pass only records its BatchNorm
        # statistics; the fake pass below splices the running-statistics update in behind its own (the reference's order: fake, then
        # real, sinskitG_model.py:1361-1374) and accumulates its gradients onto these.  VTS_D1_REAL_EARLY=0: one batched [fake | real] pass.
        p_real_early = None
        if (D1_REAL_EARLY and "D" in self.model_names and self._pair and not getattr(self.netD, "is_stylegan2_d", False)
                and getattr(self, "_I2_pyr", None) is not None and engine.PARALLEL_SCALES):
            lam = opt.lambda_G1_GAN
            pyr = [self._d1_pair(Act(S[n:2 * n]), Act(I[n:2 * n])) for S, I in zip(self._S2_pyr, self._I2_pyr)]
            in0, in1 = self._d1_pair(self._S2[n:], self._I2[n:])
            p_real_early = dict(in0=in0, in1=in1, pyr=pyr, real=True, coeff=lam, slot=slot["D_real_I"], grad_coeff=0.5 * lam,
                                stat_only=True, keep_stats=True)
            ops.step_begin(self._loss_buf, self._step_counters)      # (in front of the fork: the real pass adds into its loss slot)
            engine.msd_multi([(self.netD, [p_real_early])], self.criterionGAN, extra=lambda: self._forward_and_stacks(begin=False), extra_cost=1.0,
                             extra_main=True, streams=int(tune.get("VTS_PHASE_A_STREAMS", "3")))
        else:
            self._forward_and_stacks()
        jobs = []
        p_fake_I = p_full = None
        if "D" in self.model_names:      # compute_D1_loss
            lam = opt.lambda_G1_GAN
            if p_real_early is not None:
                pyr = self._d1_pyramid(n, pool_fake=True)
                in0, in1 = self._d1_pair(self._S2[:n], self._I2[:n])
                p_fake_I = dict(in0=in0, in1=in1, pyr=pyr, prep=self._d1_pool_prep() if pyr is not None else None,
                                groups=[dict(n0=0, n1=n, real=False, coeff=lam, slot=slot["D_fake_I"], grad_coeff=0.5 * lam)],
                                accumulate=True, ext_from=p_real_early, ext_after=0)
                jobs.append((self.netD, [p_fake_I]))
            elif getattr(self.netD, "is_stylegan2_d", False) or not self._pair:
                in0, in1 = self._d1_pair(self.real_S, self.fake_I)
                p_fake_I = dict(in0=in0, in1=in1, real=False, coeff=lam, slot=slot["D_fake_I"], grad_coeff=0.5 * lam)
                in0, in1 = self._d1_pair(self.real_S, self.real_I)
                jobs.append((self.netD, [p_fake_I, dict(in0=in0, in1=in1, real=True, coeff=lam, slot=slot["D_real_I"],
                                                        grad_coeff=0.5 * lam, accumulate=True)]))
            else:   # fake | real batched: rows [0, n) / [n, 2n) of the persistent pair buffers
                pyr = self._d1_pyramid(2 * n, pool_fake=True)
                in0, in1 = self._d1_pair(self._S2, self._I2)
                p_fake_I = dict(in0=in0, in1=in1, pyr=pyr, prep=self._d1_pool_prep() if pyr is not None else None, groups=[
                    dict(n0=0, n1=n, real=False, coeff=lam, slot=slot["D_fake_I"], grad_coeff=0.5 * lam),
                    dict(n0=n, n1=2 * n, real=True, coeff=lam, slot=slot["D_real_I"], grad_coeff=0.5 * lam)])
                jobs.append((self.netD, [p_fake_I]))
        if "D2" in self.model_names:     # compute_D2_loss
            lam2 = opt.lambda_G2_GAN
            batched = not getattr(self.netD2, "is_stylegan2_d", False)
            P = self._fake_stack.shape[0]
            passes = [] if batched else [dict(in0=self._fake_stack, real=False, coeff=lam2, slot=slot["D_fake_T_concat"], grad_coeff=0.5 * lam2)]
            # full-resolution pass: visualisation only, but it advances the BatchNorm running statistics
            # (opt.skip_D2_visualisation_pass is a measurement switch of bench.py --no_viz, not a reference option: SURVEY §8d asks
            # for the step rate with and without this pass)
            if not getattr(opt, "skip_D2_visualisation_pass", False):
                # (the sketch and mask channels of _full_stack were written by the forward's post-processing pass)
                # batched: it runs first and only records its BatchNorm statistics; the patch pass splices its running-statistics
                # update in after the fake patches, i.e. at the reference's position (sinskitG_model.py:1490-1501)
                p_full = dict(in0=self._full_stack, loss=False, stat_only=batched, pred_scales=(self.netD2.num_D - 1,))   # (only preds[-1] is shown)
                passes.append(p_full)
            K = 0
            if opt.use_more_fakeT:
                k = opt.add_fake_T_sample_size
                K = n * k
                more = self._more_stack          # built with the other stacks (_forward_and_stacks)
                if not batched:
                    passes.append(dict(in0=more, real=False, coeff=lam2, slot=slot["D_more_fake_T"], grad_coeff=0.5 * lam2, accumulate=True))
            if batched:
                groups = [dict(n0=0, n1=P, real=False, coeff=lam2, slot=slot["D_fake_T_concat"], grad_coeff=0.5 * lam2)]
                if K:
                    groups.append(dict(n0=P, n1=P + K, real=False, coeff=lam2, slot=slot["D_more_fake_T"], grad_coeff=0.5 * lam2))
                groups.append(dict(n0=P + K, n1=2 * P + K, real=True, coeff=lam2, slot=slot["D_real_T_concat"], grad_coeff=0.5 * lam2))
                passes.append(dict(in0=self._stack_all, groups=groups, ext_from=p_full if (p_full is not None) else None, ext_after=0))
            else:
                passes.append(dict(in0=self._real_stack, real=True, coeff=lam2, slot=slot["D_real_T_concat"], grad_coeff=0.5 * lam2,
                                   accumulate=True))
            jobs.append((self.netD2, passes))
        # single GPU: the generator's discriminator-free terms (_seg_g_pre: L1 / perceptual terms, patch scatter) run as one more lane
        # beside the discriminator updates instead of serially behind them (data parallel: they are their own segment, the one the
        # D / D2 all-reduces travel under)
        self._g_pre_done = False
        extra = None
        if G_PRE_LANE and not self._ddp_segments():
            def extra():
                self._seg_g_pre()
                self._g_pre_done = True
        heavy = opt.lambda_G1_lpips > 0.0 or opt.lambda_G2_lpips > 0.0      # the perceptual terms: ~90 ms of VGG convolutions
        self._chains_done = False
        if extra is not None and jobs and self._chained:
            # single GPU: each discriminator's update, its Adam step and its passes of the generator step form ONE dependency chain
            # (engine.msd_chain); the chains of D1 and D2 never wait for each other, and the generator's backward waits for D1's only
            self._run_d_chains(jobs, extra)
        else:
            engine.msd_multi(jobs, self.criterionGAN, extra=extra, extra_cost=90.0 if heavy else 0.1)
        # (visuals: the maps as the discriminator returns them -- behind its Sigmoid where gan_mode 'vanilla' gives it one)
        if p_fake_I is not None:
            self.pred_fake_I = p_fake_I["preds"][-1][:n]
            if getattr(self.netD, "use_sigmoid", False):
                self.pred_fake_I = torch.sigmoid(self.pred_fake_I)
        if p_full is not None:
            self.pred_fake_T_full = p_full["preds"][-1]
            if getattr(self.netD2, "use_sigmoid", False):
                self.pred_fake_T_full = torch.sigmoid(self.pred_fake_T_full)

    def _use_chains(self):
        """single GPU, multiscale PatchGAN discriminators, no perceptual terms: the chained schedule (engine.msd_chain / fork_lane)"""
        opt = self.opt
        heavy = opt.lambda_G1_lpips > 0.0 or opt.lambda_G2_lpips > 0.0
        nets = [getattr(self, "net" + n) for n in ("D", "D2") if n in self.model_names]
        return bool(engine.D_CHAINS and G_PRE_LANE and not self._ddp_segments() and not heavy and nets
                    and not any(getattr(net, "is_stylegan2_d", False) for net in nets)
                    and isinstance(self.netG, networks.CustomUnetGenerator))

    def _run_d_chains(self, jobs, g_pre):
        """discriminator updates + Adam(D), Adam(D2) + the discriminator passes of the generator step (what _seg_g_main does after the
        updates otherwise), one dependency chain per discriminator:
          D1: its scales on the launch stream + one side stream, the generator's L1 terms beside them, Adam(D), its pass of the
              generator step -- what the generator's backward waits for;
          D2: its update as two lanes (scale 0 | scales 1, 2) beside the WHOLE D1 chain, joined when that chain is through; then
              Adam(D2) and its forward of the generator step (a logged value: nothing of the step waits for it) as one lane that stays
              open beside the generator's backward (joined in _seg_step_chained)."""
        opt, ts, slot = self.opt, self.train_set, self._slot
        chain_d1 = chain_d2 = None
        for net, passes in jobs:
            if net is self.netD:
                def gstep_d1():
                    lam = opt.lambda_G1_GAN
                    in0, in1 = self._d1_pair(self.real_S, self.fake_I)
                    g = [dict(in0=in0, in1=in1, real=True, coeff=lam, slot=slot["G_GAN"], grad_coeff=lam, param_grads=False,
                              input_grad=(self._d_fake_I, self._have_dI), pyr=self._d1_pyramid(self.real_S.shape[0], pool_fake=False),
                              defer_merge=FUSE_MERGE)]      # the last pool^T of the pyramid merge rides in g_out_grad (_g_backward)
                    self._have_dI = True
                    self._g_gan_pass = g[0]
                    return g
                chain_d1 = dict(D=net, index0=0, update=passes, mid=lambda: self.optimizer_D.step(self._gscale, bump=False), gstep=gstep_d1)
            else:
                # G2 GAN term: fake_T_concat is detached in the reference (:1751) -> value only
                chain_d2 = dict(D=net, index0=3 if "D" in self.model_names else 0, update=passes,
                                mid=lambda: self.optimizer_D2.step(self._gscale, bump=False),
                                gstep=lambda: [dict(in0=self._fake_stack, real=True, coeff=opt.lambda_G2_GAN * ts["NT"], slot=slot["G2_GAN"])])
        self._d2_lane = None
        if chain_d2 is None or chain_d1 is None:
            engine.msd_chain(chain_d1 or chain_d2, self.criterionGAN, side=g_pre)
        elif D2_CHAIN == "serial":
            # measured (round 6): the whole D2 chain as ONE lane is ~2.5 ms of dependent small launches and becomes the step's critical
            # path (6.00 against 5.41 ms)
            self._d2_lane = engine.fork_lane(lambda: engine.msd_chain(chain_d2, self.criterionGAN, serial=True))
            engine.msd_chain(chain_d1, self.criterionGAN, side=g_pre)
        else:
            # D2's update: two lanes (scale 0 | scales 1, 2) beside the WHOLE D1 chain (update, Adam, generator-step pass); the launch
            # stream joins them when D1's chain is through, runs Adam(D2), and D2's forward of the generator step goes on as one lane
            # under the generator's backward
            D2, upd = chain_d2["D"], chain_d2["update"]
            engine._prepare_passes([(D2, upd)])

            def d2_update(scales):
                def run():
                    with ops.deferred_wgrad():
                        for sc in scales:
                            engine._scale_lane(D2, sc, upd, self.criterionGAN, knocked_out=(chain_d2["index0"] + sc) in engine.KO_LANES)
                        ops.wgrad_flush(ops.WS_LANE)
                return run
            spec = tune.get("VTS_D2_LANES", "")        # measurement: "0|1|2", "0,1,2", ... (default: scale 0 | the others)
            groups = ([[int(t) for t in g.split(",")] for g in spec.split("|")] if spec else [[0], list(range(1, D2.num_D))])
            lanes = []
            try:
                for g in groups:
                    if g:
                        lanes.append(engine.fork_lane(d2_update(g)))
                engine.msd_chain(chain_d1, self.criterionGAN, side=g_pre, serial=tune.get("VTS_D1_SERIAL", "0") == "1")
            finally:
                for h in reversed(lanes):      # (also on an exception: an open lane would keep its side stream reserved for good)
                    engine.join_lane(h)
            engine._finish_passes([(D2, upd)])
            chain_d2["mid"]()
            if tune.get("VTS_D2_TAIL", "lane") == "front":      # measurement: D2's generator-step forward as three lanes IN FRONT of the backward
                engine.msd_multi([(D2, chain_d2["gstep"]())], self.criterionGAN)
            else:
                tail = dict(chain_d2, update=[], mid=lambda: None)
                self._d2_lane = engine.fork_lane(lambda: engine.msd_chain(tail, self.criterionGAN, serial=True))
        self._chains_done = True

    def _d1_pyramid(self, rows, pool_fake):
        """input pyramid of D1 over the first `rows` samples of the pair buffers, or None (engine pools itself).  pool_fake: pool the
        fake-image rows now (the D update, right after the forward); the generator step reuses those levels."""
        if getattr(self, "_I2_pyr", None) is None or not self._pair or self.fake_I.data_ptr() != self._I2.data_ptr():
            return None
        return [self._d1_pair(Act(S[:rows]), Act(I[:rows])) for S, I in zip(self._S2_pyr, self._I2_pyr)]

    def _d1_pair(self, S, I):
        """the two concat sources of D1's first layer: (S, I) -- torch.cat((real_S, image), 1), sinskitG_model.py:1359-1372, 1671 -- or, with
        --use_cGAN False, the image alone"""
        return (S, I) if self.use_cGAN else (I, None)

    def _d1_pool_prep(self):
        """{scale: callable} pooling the fake-image rows of the D1 input pyramid INSIDE each scale's lane (engine.msd_multi `prep`):
        scale s pools its own chain from the full-resolution rows (scale 2 repeats the first level into a scratch tensor instead of
        waiting for scale 1's lane), so the pooling leaves the serial stretch between the generator forward and the lanes."""
        n = self.real_S.shape[0]
        pyr = self._I2_pyr

        def chain(s):
            def run():
                src = pyr[0][:n]
                for t in range(1, s + 1):
                    dst = pyr[t][:n] if t == s else torch.empty(n, 3, pyr[t].shape[2], pyr[t].shape[3], device=self.device)
                    ops.avgpool(src, y=dst)
                    src = dst
            return run
        return {s: chain(s) for s in range(1, len(pyr))}

    def _seg_g_pre(self):
        """the generator's loss terms that need no discriminator (compute_G1_loss / compute_G2_loss: the L1 terms).  In a data-parallel
        run this segment is what the D / D2 gradient all-reduces travel under."""
        opt, dev, ts, slot = self.opt, self.device, self.train_set, self._slot
        n, _, h, w = self.real_S.shape
        nt, P = ts["NT"], ts["real_T"].shape[0]
        self._d_fake_I = torch.empty(n, 3, h, w, device=dev)
        self._have_dI = False
        if opt.lambda_G1_L1 > 0.0:
            ops.l1(self.fake_I, self.real_I, opt.lambda_G1_L1 / self.fake_I.numel(), slot["G_L1"], self._d_fake_I, accumulate=False)
            self._have_dI = True
        if opt.lambda_G1_lpips > 0.0:
            # criterionLPIPS_vgg(fake_I, real_I).mean() * lambda (:1711)
            from vts import perceptual as P_
            P_.lpips_term(self.netLPIPS, self.fake_I, self.real_I, opt.lambda_G1_lpips / n, slot["G_lpips"], grad_into=self._d_fake_I,
                          grad_accumulate=self._have_dI)
            self._have_dI = True
        d_fake_T = None
        d_patch = None
        if opt.lambda_G2_L1 > 0.0:
            d_patch = torch.empty(P, 2, 32, 32, device=dev)
            ops.l1(self.fake_T_concat, ts["real_T"], opt.lambda_G2_L1 / (n * 2 * 32 * 32), slot["G2_L1"], d_patch)
        if opt.lambda_G2_lpips > 0.0:
            # _compute_touch_lpips_loss (:1619-1658): gx and gy as 1-channel images (the ScalingLayer broadcasts them to three channels),
            # view(-1, NT, 1, 1, 1).sum(1).mean() = sum over all patches / number of images, the two channels added
            from vts import perceptual as P_
            have = d_patch is not None
            if not have:
                d_patch = torch.empty(P, 2, 32, 32, device=dev)
            # (the two channels of the P patches as ONE batch of 2 P single-channel images -- [P, 2, 32, 32] is [2 P, 1, 32, 32] in memory --:
            #  the term is a sum over patches and channels, so one call with twice the batch replaces two; the small maps of the deep
            #  VGG layers get twice the workgroups.  VTS_LPIPS_T_SPLIT=1: one call per channel, as in round 3)
            f, r = self.fake_T_concat, ts["real_T"]
            if tune.get("VTS_LPIPS_T_SPLIT", "0") != "1" and f.is_contiguous() and r.is_contiguous() and d_patch.is_contiguous():
                P_.lpips_term(self.netLPIPS, f.view(2 * P, 1, 32, 32), r.view(2 * P, 1, 32, 32), opt.lambda_G2_lpips / n, slot["G2_lpips"],
                              grad_into=d_patch.view(2 * P, 1, 32, 32), grad_accumulate=have)
            else:
                for c in (0, 1):
                    P_.lpips_term(self.netLPIPS, f[:, c:c + 1], r[:, c:c + 1], opt.lambda_G2_lpips / n, slot["G2_lpips"],
                                  grad_into=d_patch[:, c:c + 1], grad_accumulate=have)
        if d_patch is not None:
            d_fake_T = torch.empty(n, 2, h, w, device=dev)
            ops.patch_scatter_bwd(d_patch, 0, 2, ts["offx"], ts["offy"], nt, 32, d_fake_T)
        self._d_fake_T = d_fake_T

    def _seg_g_main(self, part="all"):
        """Adam for D / D2, then the generator's GAN terms and its backward (part 'decoder': only the decoder half; the encoder half
        is _seg_g_enc, so that the decoder's gradient bucket can be all-reduced under it)"""
        opt, ts, slot = self.opt, self.train_set, self._slot
        nt = ts["NT"]
        jobs = []
        if "D" in self.model_names:
            self.optimizer_D.step(self._gscale, bump=False)
            lam = opt.lambda_G1_GAN
            in0, in1 = self._d1_pair(self.real_S, self.fake_I)
            jobs.append((self.netD, [dict(in0=in0, in1=in1, real=True, coeff=lam, slot=slot["G_GAN"], grad_coeff=lam,
                                          param_grads=False, input_grad=(self._d_fake_I, self._have_dI),
                                          pyr=self._d1_pyramid(self.real_S.shape[0], pool_fake=False), defer_merge=FUSE_MERGE)]))
            self._g_gan_pass = jobs[-1][1][0]      # (its last merge level rides in g_out_grad: _g_backward)
            self._have_dI = True
        d2_lane = None
        if "D2" in self.model_names:
            self.optimizer_D2.step(self._gscale, bump=False)
            # G2 GAN term: fake_T_concat is detached in the reference (:1751) -> value only
            g2 = dict(in0=self._fake_stack, real=True, coeff=opt.lambda_G2_GAN * nt, slot=slot["G2_GAN"])
            if D2_TAIL_LANE and engine.PARALLEL_SCALES and not getattr(self.netD2, "is_stylegan2_d", False):
                # a logged value that nothing of the step waits for: one lane that stays open under the generator's backward of this
                # segment instead of three lanes in front of it (round 6; the chained single-GPU step does the same)
                tail = dict(D=self.netD2, index0=3 if "D" in self.model_names else 0, update=[], mid=lambda: None, gstep=lambda: [g2])
                d2_lane = engine.fork_lane(lambda: engine.msd_chain(tail, self.criterionGAN, serial=True))
            else:
                jobs.append((self.netD2, [g2]))
        try:
            if jobs:
                engine.msd_multi(jobs, self.criterionGAN)
            self._g_backward(part)
        finally:
            engine.join_lane(d2_lane)

    def _seg_g_update(self):
        if getattr(self, "_chains_done", False):      # the discriminator chains already hold Adam(D / D2) and the generator step's D passes
            self._chains_done = self._g_pre_done = False
            self._g_backward("all")
            return
        if not getattr(self, "_g_pre_done", False):
            self._seg_g_pre()
        self._g_pre_done = False
        self._seg_g_main()

    def _seg_step_chained(self):
        """the whole step as ONE segment (single GPU): the D2 chain opened in _run_d_chains stays open beside the generator's backward
        and is joined in front of Adam(G)"""
        self._d2_lane = None
        try:
            self._seg_d_updates()
            self._seg_g_update()
        finally:
            engine.join_lane(self._d2_lane)
            self._d2_lane = None
        self._seg_adam_g()

    def _seg_g_enc(self):
        engine.unet_backward_encoder(self.netG, self._g_ctx, self._g_bwd_state)
        self._g_bwd_state = None

    def _g_backward(self, part="all"):
        n, _, h, w = self.real_S.shape
        d_raw = torch.empty(n, 5, h, w, device=self.device)
        gp, self._g_gan_pass = getattr(self, "_g_gan_pass", None), None
        ops.g_out_grad(self._d_fake_I if self._have_dI else None, self._d_fake_T, self.M, self.g_out, d_raw,
                       coarse=gp.get("coarse_grad") if gp is not None else None)
        if isinstance(self.netG, networks.ResnetGenerator):
            engine.resnet_backward(self.netG, self._g_ctx, d_raw)
        elif part == "decoder":
            self._g_bwd_state = engine.unet_backward_decoder(self.netG, self._g_ctx, d_raw)
        else:
            engine.unet_backward(self.netG, self._g_ctx, d_raw)

    def _seg_adam_g(self):
        self.optimizer_G.step(self._gscale, bump=False)

    def _ddp_segments(self):
        from vts import ddp as _ddp
        return bool(_ddp.active() and (self.ddp.buckets if self.ddp is not None else {}))

    def _segments(self):
        """(segment, buckets to wait for before it, buckets to start after it).  Single GPU: three segments.  Data parallel: the
        step is cut where a gradient bucket becomes complete, and every all-reduce gets compute to travel under --
          D, D2 buckets (complete after the discriminator updates)      under the generator's L1 terms (_seg_g_pre),
          G_dec (decoder gradients, complete halfway through the backward) under the encoder's backward (_seg_g_enc),
          G_enc is the exposed one (waited for right before Adam(G)).
        The reference has no counterpart (nn.DataParallel, base_model.py:104-108, reduces inside autograd)."""
        self._chained = self._use_chains()
        if self._chained:
            return [(self._seg_step_chained, (), ())]
        if not self._ddp_segments():
            return [(self._seg_d_updates, (), ("D", "D2")), (self._seg_g_update, ("D", "D2"), ("G",)), (self._seg_adam_g, ("G",), ())]
        buckets = self.ddp.buckets
        if "G_dec" in buckets:
            return [(self._seg_d_updates, (), ("D", "D2")), (self._seg_g_pre, (), ()),
                    (lambda: self._seg_g_main("decoder"), ("D", "D2"), ("G_dec",)), (self._seg_g_enc, (), ("G_enc",)),
                    (self._seg_adam_g, ("G_dec", "G_enc"), ())]
        return [(self._seg_d_updates, (), ("D", "D2")), (self._seg_g_pre, (), ()), (self._seg_g_main, ("D", "D2"), ("G",)),
                (self._seg_adam_g, ("G",), ())]

    def _comm(self, name, start):
        if self.ddp is None or name not in self.ddp.buckets:
            return
        b = self.ddp.buckets[name]
        b.start() if start else b.wait()

    def _drop_graphs(self):
        self._infer_graph, self._infer_eager_done = None, False
        ops.release_ws((id(self), "infer"))
        if getattr(self, "_graphs", None) is not None:
            self._graphs = None
            ops.release_ws((id(self), "train"))

    _INFER_OUTPUTS = ("g_out", "fake_I", "fake_T", "fake_N", "fake_gx", "fake_gy", "aug_fake_I", "aug_real_I", "_full_stack", "_g_ctx")

    _STEP_OUTPUTS = ("g_out", "fake_I", "fake_T", "fake_N", "fake_gx", "fake_gy", "aug_fake_I", "aug_real_I", "_full_stack", "_stack_all",
                     "_fake_stack", "_real_stack", "_more_stack", "fake_T_concat", "pred_fake_I", "pred_fake_T_full", "_g_ctx",
                     "fake_sample_offset_x", "fake_sample_offset_y")

    def _capture_graphs(self):
        """Capture the segments as HIP graphs sharing one memory pool (torch.cuda.CUDAGraph over
        the launch stream our ctypes kernels use).  Capturing records work without executing it."""
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
                    nodes.append(ops.capture_node_count())      # (all nodes, kernel nodes) of this segment: the replayed launch count
                graphs.append(g)
        except Exception:
            ops.release_ws((id(self), "train"))
            raise
        for o, c in zip(self.optimizers, counts):
            o.step_count = c   # host mirrors moved during capture; the device counters did not
        self._graphs = graphs
        self.graph_nodes = nodes
        # the tensors the captured step writes (visuals, predictions, patch stacks): a validation test() in between rebinds these
        # attributes to ITS outputs, so every replay re-attaches the training ones (get_current_visuals / compute_metrics('train_'))
        self._graph_attrs = {k: getattr(self, k) for k in self._STEP_OUTPUTS if hasattr(self, k)}

    def optimize_parameters(self, epoch=0, timing=False):
        if self.train_set is None:
            raise RuntimeError("optimize_parameters needs tactile patches in the batch (T_images)")
        try:
            self._check_unbuilt_terms(self.opt, epoch)
        except NotImplementedError:
            # the run ends here: keep what was trained since the last periodic checkpoint (--save_epoch_freq 50 by default)
            if not getattr(self, "_saved_before_stop", False):
                self._saved_before_stop = True
                try:
                    self.save_networks("latest")
                except OSError as e:
                    print("could not write the 'latest' checkpoint before stopping: %s" % e, flush=True)
            raise
        self._gscale = self.ddp.grad_scale if self.ddp is not None else 1.0
        for o in self.optimizers:
            o.sync_lr()
        self._prepare_ranks()
        use_graph = bool(getattr(self.opt, "use_hip_graph", False)) and self._draws is None
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
            self.__dict__.update(self._graph_attrs)
            for o in self.optimizers:
                o.step_count += 1
        else:
            self._eager_steps_done += 1
            self._g_ctx = None if not use_graph else self._g_ctx

    # ------------------------------------------------------------------ evaluation metrics
    METRICS = ("I_PSNR", "T_AE", "T_MSE", "I_SSIM")

    def _sifid_net(self):
        """Inception block 0 for I_SIFID / T_SIFID, or None: built when --inception_weights (a torchvision / pytorch-fid state dict) is
        given, or with VTS_SIFID=1 on the seeded stand-in weights (values then only compare builds on the same seed)."""
        if not hasattr(self, "_inception"):
            self._inception = None
            if getattr(self.opt, "inception_weights", "") or os.environ.get("VTS_SIFID", "0") == "1":
                from . import inception
                self._inception = inception.build(self.opt, self.device)
        return self._inception

    def compute_metrics(self, prefix=""):
        """Evaluation metrics of the current outputs (reference: compute_evaluation_metric, models/model_utils.py:431-561, called from
        compute_visuals sinskitG_model.py:889-925): I_PSNR, I_SSIM, T_AE, T_MSE on the validation patches (the training patches with
        prefix 'train_'), I_SIFID / T_SIFID when an Inception block is available (_sifid_net), I_LPIPS / T_LPIPS when the LPIPS network is
        (it is whenever an LPIPS loss term is on, with --lpips_weights, or with VTS_LPIPS_METRICS=1 on the stand-in weights)."""
        pset = self.train_set if prefix == "train_" else self.val_set
        if pset is None or not hasattr(self, "real_I") or self.test_edit_S:
            return {}
        P = pset["real_T"].shape[0]
        fake_T_concat = torch.empty(P, 2, 32, 32, device=self.device)
        self._gather(self.fake_T, pset, fake_T_concat, 0, channels=2)
        vals = ops.eval_metrics(self.real_I, self.fake_I, pset["real_T"], fake_T_concat).cpu().tolist()
        names = list(self.METRICS)
        net = self._sifid_net()
        if net is not None:
            names += ["I_SIFID", "T_SIFID"]
            vals += [float(engine.sifid_images(net, self.real_I.contiguous(), self.fake_I.contiguous())),
                     float(engine.sifid_tactile(net, pset["real_T"], fake_T_concat))]
            self.metric_sifid_pretrained = bool(net.pretrained)
        alex_phase = not self.isTrain        # eval_LPIPS: VGG while training / validating (:497-499), AlexNet in the test phase (:501)
        want_lpips = (getattr(self.opt, "lpips_alex_weights", "") if alex_phase else getattr(self.opt, "lpips_weights", "")) or \
            os.environ.get("VTS_LPIPS_METRICS", "0") == "1" or (self.netLPIPS is not None and not alex_phase)
        if want_lpips:
            # I_LPIPS / T_LPIPS (model_utils.py:475-478, 521-527) with the reference's backbone of the phase
            from vts import perceptual as P_
            buf = ops.loss_slots(2, self.device)
            n_img = self.real_I.shape[0]
            if alex_phase:
                net = self._lpips_alex_net()
                term = lambda a, b, coeff, slot: P_.lpips_alex_value(net, a, b, coeff, slot)      # noqa: E731
            else:
                net = self._lpips_net()
                term = lambda a, b, coeff, slot: P_.lpips_term(net, a, b, coeff, slot)            # noqa: E731
            term(self.real_I.contiguous(), self.fake_I.contiguous(), 1.0 / n_img, buf[0:1])
            for c in (0, 1):     # nearest resize to 224 x 224, fake clamped to [0, 1], each channel tiled to three; mean over patches, gx + gy
                a = ops.sifid_input(pset["real_T"], c, 1, size=(224, 224))
                b = ops.sifid_input(fake_T_concat, c, 1, size=(224, 224), clamp01=True)
                for i0 in range(0, P, 32):
                    term(a[i0:i0 + 32], b[i0:i0 + 32], 1.0 / P, buf[1:2])
            lv = ops.loss_values(buf)
            names += ["I_LPIPS", "T_LPIPS"]
            vals += [lv[0], lv[1]]
            self.metric_lpips_pretrained, self.metric_lpips_backbone = bool(net.pretrained), "alex" if alex_phase else "vgg"
        # I_SSIM: torchmetrics is an unpinned pip dependency that cannot be run here -- the kernel is pinned to a restatement of its
        # published algorithm (oracle/nets.py:ssim, cross-checked against an independent float64 evaluation in tests/test_oracle_golden.py)
        self.metric_ssim_pinned = "restatement"
        for name, v in zip(names, vals):
            setattr(self, "metric_%s%s" % (prefix, name), v)
            if prefix + name not in self.metric_names:
                self.metric_names.append(prefix + name)
        return {prefix + n: v for n, v in zip(names, vals)}

    # ------------------------------------------------------------------ logging
    def get_current_losses(self):
        vals = ops.loss_values(self._loss_buf)   # the only device->host sync of the loss path
        self._check_candidate_counts(wait=True)
        for i, name in enumerate(LOSS_SLOTS):
            setattr(self, "loss_" + name, vals[i])
        return BaseModel.get_current_losses(self)

    def compute_visuals(self):
        pass

    def get_current_visuals(self):
        return BaseModel.get_current_visuals(self)
