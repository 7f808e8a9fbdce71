"""Drop-in for the reference's ``networks.osvos_vgg.OSVOS_VGG`` (src/networks/osvos_vgg.py:17-153).

Same constructor, same sub-module attribute names (stages, side_prep, score_dsn, upscale, upscale_,
fuse), same 52-tensor state_dict (keys, shapes, order) and the same ``forward(x) -> list of 5
[N,1,H,W] logit maps``.  The sub-modules are stock torch.nn containers that only HOLD the fp32
parameters (so optimizers, checkpoints and code that walks ``net.stages`` keep working); the
arithmetic of forward/backward runs in the hand-written HIP kernels through fosvos_hip.engine.
"""
import os
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.modules as modules

from fosvos_hip import engine
from layers.osvos_layers import interp_surgery
from util.logger import get_logger

log = get_logger(__file__)


class OSVOS_VGG(nn.Module):
    def __init__(self, pretrained=1):
        super(OSVOS_VGG, self).__init__()
        lay_list = [[64, 64],
                    ['M', 128, 128],
                    ['M', 256, 256, 256],
                    ['M', 512, 512, 512],
                    ['M', 512, 512, 512]]
        in_channels = [3, 64, 128, 256, 512]

        log.info("Constructing OSVOS architecture...")
        stages = modules.ModuleList()
        side_prep = modules.ModuleList()
        score_dsn = modules.ModuleList()
        upscale = modules.ModuleList()
        upscale_ = modules.ModuleList()
        for i, cfg in enumerate(lay_list):
            stages.append(self._make_layers_osvos(cfg, in_channels[i]))
            if i > 0:  # side branches hang off stages 2..5
                side_prep.append(nn.Conv2d(cfg[-1], 16, kernel_size=3, padding=1))
                score_dsn.append(nn.Conv2d(16, 1, kernel_size=1, padding=0))
                upscale_.append(nn.ConvTranspose2d(1, 1, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))
                upscale.append(nn.ConvTranspose2d(16, 16, kernel_size=2 ** (1 + i), stride=2 ** i, bias=False))

        # registration order fixes the state_dict order
        self.upscale = upscale
        self.upscale_ = upscale_
        self.stages = stages
        self.side_prep = side_prep
        self.score_dsn = score_dsn
        self.fuse = nn.Conv2d(64, 1, kernel_size=1, padding=0)

        self._packs = engine.PackedWeights()  # bf16 MFMA images of the weights, rebuilt when a master changes
        # Opt-in for training loops that call ``loss.backward()`` (never ``torch.autograd.grad``): let the wgrad
        # kernels add straight into existing ``p.grad`` tensors instead of handing autograd a fresh tensor to add.
        self.accumulate_grads_in_place = False
        # The four side logit maps (score_dsn -> upscale_ -> crop) are outputs of the reference's forward; the online
        # loop only reads outputs[-1].  False skips them in the fused head (half of its arithmetic and 4 x 1.6 MB of
        # writes per frame): forward then returns four EMPTY placeholder tensors in their place.
        self.compute_side_outputs = True

        log.info("Initializing weights")
        self._initialize_weights(pretrained)

    # ------------------------------------------------------------------ forward on the HIP kernels
    def forward(self, x):
        """list of 5 logit maps [N,1,H,W]: the 4 side outputs then the fused output."""
        params = self._ordered_params()
        return engine.run(self._packs, params, x, with_side_out=bool(getattr(self, 'compute_side_outputs', True)),
                          inplace_grad=getattr(self, 'accumulate_grads_in_place', False))

    def prepack_weights(self, prefixes=None):
        """Rebuild NOW (on the current stream) the bf16 MFMA images of the 3x3 conv weights whose fp32 masters changed since
        they were last packed - all of them, or those whose state_dict name starts with one of `prefixes`.  `forward` does
        this for every stale layer by itself; a loop that steps part of the model early can repack that part early too."""
        P = dict(zip(engine.PARAM_NAMES, self._ordered_params()))
        names = [n for n in engine.PARAM_NAMES if n.endswith(".weight") and n.startswith(("stages.", "side_prep."))
                 and n != engine._CONV_NAMES[0][0]]  # (not conv1_1: its kernel reads the fp32 master)
        if prefixes is not None:
            names = [n for n in names if n.startswith(tuple(prefixes))]
        if names:
            self._packs.conv_many([(n, P[n]) for n in names])

    def join_gradients(self):
        """With ``defer_wgrad_join`` the weight-gradient kernels of a backward pass may still be running on the
        auxiliary stream when ``backward()`` returns; call this before reading ``p.grad`` (optimizer step,
        all-reduce).  No-op otherwise."""
        self._packs.arenas.join()

    @property
    def defer_wgrad_join(self):
        return getattr(self._packs, "defer_wgrad_join", False)

    @defer_wgrad_join.setter
    def defer_wgrad_join(self, value):
        if not value:
            self._packs.arenas.join()
        self._packs.defer_wgrad_join = bool(value)

    @property
    def forward_one_stream(self):
        return getattr(self._packs, "forward_one_stream", False)

    @forward_one_stream.setter
    def forward_one_stream(self, value):
        """True: a batched forward pass keeps both of its chains of frames on the caller's stream (the online loop sets it
        while the passes of a cycle alternate between two streams of their own)."""
        self._packs.forward_one_stream = bool(value)

    @property
    def publish_grad_buckets(self):
        return getattr(self._packs, "publish_grad_buckets", False)

    @publish_grad_buckets.setter
    def publish_grad_buckets(self, value):
        """True: backward passes publish their gradients in completion order (stage 5, stage 4, stage 3, the rest) through
        ``wait_grad_bucket`` - what the data-parallel loops overlap their bucketed all-reduce with."""
        self._packs.publish_grad_buckets = bool(value)

    @property
    def publishes_grad_buckets(self):
        """Whether backward passes of this module can publish gradient buckets at all: only the native layer loop records
        the bucket events (the per-op Python engine behind FOSVOS_PY_ENGINE=1 does not)."""
        return bool(engine.USE_NATIVE_LOOP)

    @property
    def reserve_arena_frames(self):
        return self._packs.arenas.reserve_frames

    @reserve_arena_frames.setter
    def reserve_arena_frames(self, value):
        """The largest batch of one frame size the caller's loop will pass: activation arenas allocated from now on are
        sized for every batch up to it (engine.ArenaPool.reserve_frames), one allocation per frame size."""
        self._packs.arenas.reserve_frames = max(0, int(value))

    @property
    def overwrite_grads(self):
        return getattr(self._packs, "overwrite_grads", False)

    @overwrite_grads.setter
    def overwrite_grads(self, value):
        """True: the next backward passes WRITE the parameter gradients they compute (all but score_dsn's, which only a loss
        on the side outputs produces - such a pass refuses the flag) instead of adding them to ``p.grad``: what an
        accumulation cycle's first pass may do when nothing else adds to the buffers beside it, so that the buffers need no
        zeroing between cycles.  The online loop sets it around a cycle that is one batched pass."""
        self._packs.overwrite_grads = bool(value)

    @property
    def last_pass_of_cycle(self):
        return getattr(self._packs, "last_pass_of_cycle", False)

    @last_pass_of_cycle.setter
    def last_pass_of_cycle(self, value):
        """Hint for the next backward pass: no forward pass follows it before the optimizer step, so its trailing
        weight-gradient kernels may take the whole chip (the training loops set it around a cycle's last pass)."""
        self._packs.last_pass_of_cycle = bool(value)

    def wait_grad_bucket(self, bucket, stream=None):
        """Make `stream` (default: the current one) wait for gradient bucket `bucket` (``parallel.VGG_BUCKETS`` order) of
        the last backward pass OF THIS MODULE run with ``publish_grad_buckets`` (the events live in the module's own
        fosvos_ctx, so another model training on the same device never interferes); returns at once on the host."""
        import torch
        from fosvos_hip import check, lib
        dev = next(self.parameters()).device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        st = stream if stream is not None else torch.cuda.current_stream(idx)
        check(lib().fosvos_vgg_grad_bucket_wait(self._packs.arenas.ctx(idx), int(bucket), st.cuda_stream), "vgg_grad_bucket_wait")

    def _ordered_params(self):
        """The 52 parameters in state_dict order, by direct attribute access (named_parameters() walks the whole
        module tree: 0.15 ms per call)."""
        ps = [m.weight for m in self.upscale] + [m.weight for m in self.upscale_]
        for stage in self.stages:
            for m in stage:
                if isinstance(m, nn.Conv2d):
                    ps += [m.weight, m.bias]
        for mods in (self.side_prep, self.score_dsn):
            for m in mods:
                ps += [m.weight, m.bias]
        ps += [self.fuse.weight, self.fuse.bias]
        return ps

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_packs', None)  # whole-module pickles (NetworkProvider.save_model) carry no device caches
        state.pop('_fosvos_flat_grads', None)  # ... nor a training loop's flat gradient buffer (parallel.FlatGrads.attach)
        state['compute_side_outputs'] = True        # ... and no loop-local switches (a snapshot taken inside _train)
        state['accumulate_grads_in_place'] = False
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._packs = engine.PackedWeights()

    # ------------------------------------------------------------------ structure / init (host side)
    @staticmethod
    def _make_layers_osvos(cfg, in_channels):
        layers = []
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))
            else:
                layers.extend([nn.Conv2d(in_channels, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)])
                in_channels = v
        return nn.Sequential(*layers)

    def _initialize_weights(self, pretrained):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.001)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.ConvTranspose2d):
                m.weight.data.zero_()
                m.weight.data = interp_surgery(m)

        if pretrained == 1:
            self._load_from_pytorch()
        elif pretrained == 2:
            self._load_from_caffe()

    def _load_from_pytorch(self):
        log.info('Loading weights from PyTorch VGG')
        try:
            from torchvision.models import vgg16
        except ImportError as e:  # torchvision is not part of this image
            raise RuntimeError("OSVOS_VGG(pretrained=1) copies torchvision's ImageNet VGG-16, which needs "
                               "torchvision and network access; use pretrained=0 + load_state_dict, or "
                               "pretrained=2 with vgg_hed_caffe.mat") from e
        _vgg = vgg16(pretrained=True)
        src = [m for m in _vgg.features if isinstance(m, nn.Conv2d)]
        dst = [m for stage in self.stages for m in stage if isinstance(m, nn.Conv2d)]
        for d, s in zip(dst, src):
            d.weight = deepcopy(s.weight)
            d.bias = deepcopy(s.bias)

    def _load_from_caffe(self):
        log.info('Loading weights from Caffe VGG')
        import scipy.io
        from config.mypath import Path
        caffe_weights = scipy.io.loadmat(os.path.join(Path.models_dir(), 'vgg_hed_caffe.mat'))
        convs = [m for stage in self.stages for m in stage if isinstance(m, nn.Conv2d)]
        for k, conv in enumerate(convs):
            c_w = torch.from_numpy(caffe_weights['weights'][0][k].transpose())
            c_b = torch.from_numpy(caffe_weights['biases'][0][k][:, 0])
            assert conv.weight.data.shape == c_w.shape
            assert conv.bias.data.shape == c_b.shape
            conv.weight.data = c_w
            conv.bias.data = c_b
