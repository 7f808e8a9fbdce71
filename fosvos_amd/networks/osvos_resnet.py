"""OSVOS_RESNET - same constructor, module tree and state_dict keys as the reference class
(src/networks/osvos_resnet.py:15-150), inference on the hand-written HIP kernels of csrc/resnet.hip.

This is the reference's distillation student (src/mimic.py:70) and the net its filter pruning rewrites
(src/prune.py:297-481): ResNet-18/34/50/101/152 trunk with ``scale_down_exponent`` channel thinning, a 3x3 side_prep
conv per stage and the OSVOS side-output / fuse head with strides 4, 8, 16, 32.  The reference builds the trunk from
torchvision's BasicBlock / Bottleneck; torchvision is not part of this image, so the two blocks are restated here with
the same attribute names (conv1, bn1, relu, conv2, bn2[, conv3, bn3], downsample, stride) and therefore the same keys.

Scope: eval-mode forward only.  BatchNorm uses its running statistics and is folded into the conv in
front of it when the weights are packed (fosvos_hip/resnet_engine.py); a forward in training mode raises instead of
falling back to anything.
"""
import torch.nn as nn

from fosvos_hip import resnet_engine
from layers.osvos_layers import interp_surgery
from util.logger import get_logger

log = get_logger(__file__)


def _block_only_message(name):
    return ("%s holds parameters only: it runs inside OSVOS_RESNET.forward, where the HIP path fuses conv, folded "
            "BatchNorm, residual add and ReLU" % name)


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock: conv3x3(stride) - bn - relu - conv3x3 - bn, + identity/downsample, relu."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError(_block_only_message("BasicBlock"))


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck: 1x1 - 3x3(stride) - 1x1 (x4 planes), each with bn, + residual, relu."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError(_block_only_message("Bottleneck"))


class OSVOS_RESNET(nn.Module):
    def __init__(self, pretrained, version=18, n_channels_input=3, n_channels_output=1, scale_down_exponent=0,
                 is_mode_mimic=False):
        self.is_mode_mimic = is_mode_mimic
        self.scale_down_exponent = scale_down_exponent
        self.inplanes = 64 // (2 ** scale_down_exponent)
        super(OSVOS_RESNET, self).__init__()
        log.info("Constructing OSVOS resnet architecture...")

        block, layers = self._match_version(version)
        n_channels_side_inputs = [c // (2 ** scale_down_exponent) for c in (64, 128, 256, 512)]
        self.layer_base = self._make_layer_base(n_channels_input, n_channels_side_inputs[0])
        self.layer_stages = self._make_layer_stages(block, layers, n_channels_side_inputs)
        # (as in the reference the side branches are sized for BasicBlock trunks: with the Bottleneck versions the
        # stage outputs are 4x wider than side_prep expects and forward raises, here as there)
        (self.side_prep, self.upscale_side_prep, self.score_dsn, self.upscale_score_dsn,
         self.layer_fuse) = self._make_osvos_layers(n_channels_side_inputs, n_channels_output)
        self._plan = resnet_engine.ResnetPlan()
        # True (the reference's contract): forward returns the 4 side logit maps and the fused one.  A caller that reads
        # outputs[-1] only may set it to False: the head then skips the side maps (empty placeholders are returned).
        self.compute_side_outputs = True
        self._initialize_weights()
        if pretrained:
            self._load_from_pytorch(version)

    # ------------------------------------------------------------------ forward on the HIP kernels
    def forward(self, x):
        """list of 5 logit maps [N,1,H,W]: the 4 side outputs, then the fused output
        (src/networks/osvos_resnet.py:42-68)."""
        return resnet_engine.forward(self, self._plan, x)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_plan', None)
        state['compute_side_outputs'] = True
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._plan = resnet_engine.ResnetPlan()

    # ------------------------------------------------------------------ structure / init (host side)
    @staticmethod
    def _match_version(version):
        table = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
                 101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}
        if version not in table:
            raise Exception('Invalid version for resnet. Must be one of [18, 34, 50, 101, 152].')
        return table[version]

    @staticmethod
    def _make_layer_base(n_channels_input, n_channels_output):
        return nn.Sequential(nn.Conv2d(n_channels_input, n_channels_output, kernel_size=7, stride=2, padding=3, bias=False),
                             nn.BatchNorm2d(n_channels_output), nn.ReLU(inplace=True),
                             nn.MaxPool2d(kernel_size=3, stride=2, padding=1))

    def _make_layer_stages(self, block, layers, n_channels_side_inputs):
        return nn.ModuleList([self._make_layer(block, n_channels_side_inputs[0], layers[0]),
                              self._make_layer(block, n_channels_side_inputs[1], layers[1], stride=2),
                              self._make_layer(block, n_channels_side_inputs[2], layers[2], stride=2),
                              self._make_layer(block, n_channels_side_inputs[3], layers[3], stride=2)])

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                                                 bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    @staticmethod
    def _make_osvos_layers(channels_side_input, n_channels_output, n_channels_output_side_prep=16,
                           n_channels_output_upscale_side_prep=16):
        side_prep, upscale_side_prep = nn.ModuleList(), nn.ModuleList()
        score_dsn, upscale_score_dsn = nn.ModuleList(), nn.ModuleList()
        for index, n_channels in enumerate(channels_side_input):
            side_prep.append(nn.Conv2d(n_channels, n_channels_output_side_prep, kernel_size=3, padding=1))
            upscale_side_prep.append(nn.ConvTranspose2d(n_channels_output_side_prep, n_channels_output_upscale_side_prep,
                                                        kernel_size=2 ** (3 + index), stride=2 ** (2 + index), bias=False))
            score_dsn.append(nn.Conv2d(n_channels_output_side_prep, n_channels_output, kernel_size=1, padding=0))
            upscale_score_dsn.append(nn.ConvTranspose2d(n_channels_output, n_channels_output,
                                                        kernel_size=2 ** (3 + index), stride=2 ** (2 + index), bias=False))
        layer_fuse = nn.Conv2d(n_channels_output_upscale_side_prep * len(channels_side_input), n_channels_output,
                               kernel_size=1, padding=0)
        return side_prep, upscale_side_prep, score_dsn, upscale_score_dsn, layer_fuse

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.001)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.ConvTranspose2d):
                m.weight.data.zero_()
                m.weight.data = interp_surgery(m)

    def _load_from_pytorch(self, version):
        log.info('Loading weights from PyTorch Resnet')
        try:
            import torchvision.models as tvm
        except ImportError as e:  # torchvision is not part of this image
            raise RuntimeError("OSVOS_RESNET(pretrained=True) copies torchvision's ImageNet ResNet-%d, which needs "
                               "torchvision and network access; use pretrained=False + load_state_dict" % version) from e
        resnet = getattr(tvm, 'resnet%d' % version)(pretrained=True)
        want = self.state_dict()
        got = {}
        for k, v in resnet.state_dict().items():
            if k.startswith('conv1.'):
                got['layer_base.0.' + k[6:]] = v
            elif k.startswith('bn1.'):
                got['layer_base.1.' + k[4:]] = v
            elif k.startswith('layer'):
                stage, rest = k.split('.', 1)
                got['layer_stages.%d.%s' % (int(stage[5:]) - 1, rest)] = v
        for k, v in got.items():
            if k in want and want[k].shape == v.shape:  # (a thinned net keeps its own, narrower tensors)
                want[k].copy_(v)


class BasicBlockDummy(nn.Module):
    """The block src/prune.py rebuilds around pruned convs (src/networks/osvos_resnet.py:187-216): same attribute
    names as BasicBlock, so the native path treats it the same."""
    expansion = 1

    def __init__(self, conv1, bn1, relu, conv2, bn2, downsample, stride):
        super(BasicBlockDummy, self).__init__()
        self.conv1 = conv1
        self.bn1 = bn1
        self.relu = relu
        self.conv2 = conv2
        self.bn2 = bn2
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError(_block_only_message("BasicBlockDummy"))
