"""Oracle networks: torch-CPU restatement of smp 0.3.3 Unet / UnetPlusPlus /
Linknet / FPN over torchvision ResNet encoders (TEST INFRASTRUCTURE ONLY).

The module tree reproduces the upstream attribute names so that
``state_dict()`` keys equal the ones a reference checkpoint holds
(``model.encoder.layer1.0.conv1.weight`` ...).
Call site restated: reference ``src/models/smp/model.py:38-44``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# smp.encoders.get_preprocessing_params(name) for the torchvision ResNets
# (reference src/models/smp/model.py:49).
_IMAGENET = {
    'input_space': 'RGB',
    'input_range': [0, 1],
    'mean': [0.485, 0.456, 0.406],
    'std': [0.229, 0.224, 0.225],
}

ENCODER_CHANNELS = {
    'resnet18': (3, 64, 64, 128, 256, 512),
    'resnet34': (3, 64, 64, 128, 256, 512),
    'resnet50': (3, 64, 256, 512, 1024, 2048),
    'resnet101': (3, 64, 256, 512, 1024, 2048),
    'resnet152': (3, 64, 256, 512, 1024, 2048),
}
_RESNET_CFG = {
    'resnet18': ('basic', (2, 2, 2, 2)),
    'resnet34': ('basic', (3, 4, 6, 3)),
    'resnet50': ('bottleneck', (3, 4, 6, 3)),
    'resnet101': ('bottleneck', (3, 4, 23, 3)),
    'resnet152': ('bottleneck', (3, 8, 36, 3)),
}


def get_preprocessing_params(encoder_name, pretrained='imagenet'):
    if encoder_name not in ENCODER_CHANNELS:
        raise KeyError(f'Wrong encoder name `{encoder_name}`')
    return dict(_IMAGENET)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNetEncoder(nn.Module):
    """torchvision ResNet minus avgpool/fc, returning the 6 smp features."""

    def __init__(self, name, in_channels=3, depth=5):
        super().__init__()
        self._depth = depth            # smp get_encoder(depth=...): the stages stay in the module (and in state_dict), forward stops early
        kind, layers = _RESNET_CFG[name]
        block = BasicBlock if kind == 'basic' else Bottleneck
        self.out_channels = ENCODER_CHANNELS[name][:depth + 1]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        for m in self.modules():  # torchvision init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def make_dilated(self, output_stride):
        """smp EncoderMixin.make_dilated (encoders/_base.py) + utils.replace_strides_with_dilation: output_stride 16 turns EVERY conv of
        layer4 into stride 1 / dilation 2 / padding (k // 2) * 2 (the 1x1 convs and the downsample included), 8 also layer3 with 2 and
        layer4 with 4."""
        if output_stride == 16:
            stages = [(self.layer4, 2)]
        elif output_stride == 8:
            stages = [(self.layer3, 2), (self.layer4, 4)]
        else:
            raise ValueError(f'Output stride should be 16 or 8, got {output_stride}.')
        for stage, rate in stages:
            for m in stage.modules():
                if isinstance(m, nn.Conv2d):
                    m.stride = (1, 1)
                    m.dilation = (rate, rate)
                    kh, kw = m.kernel_size
                    m.padding = ((kh // 2) * rate, (kw // 2) * rate)

    def forward(self, x):
        feats = [x, self.relu(self.bn1(self.conv1(x)))]
        stages = [lambda t: self.layer1(self.maxpool(t)), self.layer2, self.layer3, self.layer4]
        for stage in stages[:self._depth - 1]:
            feats.append(stage(feats[-1]))
        return feats



# ------------------------------------------------------------------------------------------------ RegNet encoders
# smp 0.3.3 ``encoders/timm_regnet.py`` (RegNetEncoder over timm==0.9.2 ``models/regnet.py``), the ``timm-regnetx_002`` /
# ``timm-regnetx_064`` / ``timm-regnety_120`` rows of the reference's sweep (configs/tune.yaml:19-24).  Neither package is installed
# here: restated from the published sources -- stage widths from timm's generate_regnet / adjust_widths_groups_comp, module tree and
# attribute names (stem.conv / stem.bn, s{i}.b{j}.conv{1,2,3}.{conv,bn}, .se.fc{1,2}, .downsample.{conv,bn}) as timm's, so that
# state_dict keys equal a reference checkpoint's.  Pinned by the published parameter counts (tests/test_oracle.py).
_REGNET_CFG = {   # smp's _mcfg rows: w0, wa, wm, group_w, depth, se_ratio (bottle_ratio 1.0, stem_width 32)
    'timm-regnetx_002': dict(w0=24, wa=36.44, wm=2.49, group_w=8, depth=13, se_ratio=0.0),
    'timm-regnetx_064': dict(w0=184, wa=60.83, wm=2.07, group_w=56, depth=17, se_ratio=0.0),
    'timm-regnety_120': dict(w0=168, wa=73.36, wm=2.37, group_w=112, depth=19, se_ratio=0.25),
}


def regnet_stages(cfg):
    """(widths, depths, group widths) of the four stages: timm generate_regnet (quant 8) + adjust_widths_groups_comp (bottle_ratio 1)."""
    import numpy as np
    cont = np.arange(cfg['depth']) * cfg['wa'] + cfg['w0']
    exps = np.round(np.log(cont / cfg['w0']) / np.log(cfg['wm']))
    widths = (np.round(np.divide(cfg['w0'] * np.power(cfg['wm'], exps), 8)) * 8).astype(int).tolist()
    stage_w, stage_d = [], []
    for w in widths:
        if stage_w and stage_w[-1] == w:
            stage_d[-1] += 1
        else:
            stage_w.append(w); stage_d.append(1)
    gs = [min(cfg['group_w'], w) for w in stage_w]
    stage_w = [int(round(w / g) * g) for w, g in zip(stage_w, gs)]       # bottleneck width a multiple of the group width
    return stage_w, stage_d, gs


for _n, _c in _REGNET_CFG.items():
    ENCODER_CHANNELS[_n] = (3, 32) + tuple(regnet_stages(_c)[0])


class ConvNormAct(nn.Module):
    """timm.layers.ConvNormAct: .conv (no bias) + .bn (BatchNormAct2d: BatchNorm2d with the activation fused behind it)."""

    def __init__(self, cin, cout, k, stride=1, groups=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.act else x


class SEModule(nn.Module):
    """timm.layers.SEModule: mean over (H, W) -> fc1 (1x1 conv, bias) -> ReLU -> fc2 -> sigmoid gate."""

    def __init__(self, channels, rd_channels):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, rd_channels, 1, bias=True)
        self.fc2 = nn.Conv2d(rd_channels, channels, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = self.fc2(F.relu(self.fc1(s)))
        return x * s.sigmoid()


class RegNetBottleneck(nn.Module):
    """timm regnet.Bottleneck (bottle_ratio 1): conv1 1x1, conv2 grouped 3x3 (stride), optional SE with in_chs * se_ratio reduction
    channels, conv3 1x1 without activation, 1x1 conv shortcut where the shape changes, ReLU behind the sum."""

    def __init__(self, cin, cout, stride, group_w, se_ratio):
        super().__init__()
        self.conv1 = ConvNormAct(cin, cout, 1)
        self.conv2 = ConvNormAct(cout, cout, 3, stride, groups=cout // group_w)
        self.se = SEModule(cout, int(round(cin * se_ratio))) if se_ratio else None
        self.conv3 = ConvNormAct(cout, cout, 1, act=False)
        self.downsample = ConvNormAct(cin, cout, 1, stride, act=False) if (cin != cout or stride != 1) else None

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        if self.se is not None:
            y = self.se(y)
        y = self.conv3(y)
        return F.relu(y + (self.downsample(x) if self.downsample is not None else x))


class RegNetEncoder(nn.Module):
    """smp RegNetEncoder.get_stages: [identity, stem, s1, s2, s3, s4] (head deleted; final_conv is an Identity for these configs)."""

    def __init__(self, name, in_channels=3, depth=5):
        super().__init__()
        cfg = _REGNET_CFG[name]
        self._depth = depth
        self.out_channels = ENCODER_CHANNELS[name][:depth + 1]
        widths, depths, gs = regnet_stages(cfg)
        self.stem = ConvNormAct(in_channels, 32, 3, 2)
        prev = 32
        for i, (w, d, g) in enumerate(zip(widths, depths, gs)):
            stage = nn.Sequential()
            for j in range(d):
                stage.add_module(f'b{j + 1}', RegNetBottleneck(prev, w, 2 if j == 0 else 1, g, cfg['se_ratio']))
                prev = w
            self.add_module(f's{i + 1}', stage)
        for m in self.modules():      # timm regnet._init_weights (zero_init_last=True: conv3.bn.weight = 0)
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    m.bias.data.zero_()
        for m in self.modules():
            if isinstance(m, RegNetBottleneck):
                nn.init.zeros_(m.conv3.bn.weight)

    def forward(self, x):
        feats = [x]
        x = self.stem(x); feats.append(x)
        for i in range(1, self._depth):
            x = getattr(self, f's{i}')(x); feats.append(x)
        return feats



# ------------------------------------------------------------------------------------------------ EfficientNet encoders
# smp 0.3.3 ``encoders/efficientnet.py`` (EfficientNetEncoder over efficientnet_pytorch 0.7.1's ``EfficientNet``), the ``efficientnet-b0 / b5 /
# b7`` rows of the reference's sweep (configs/tune.yaml:25-28).  Neither package is installed here: restated from the published sources --
# block arguments, round_filters / round_repeats, Conv2dStaticSamePadding (TF "same" padding FIXED at construction from the model's nominal
# image size: 224 / 456 / 600), MBConvBlock (expand 1x1 -> depthwise k3 / k5 -> squeeze-excite with swish -> project 1x1, id skip with
# drop_connect), BatchNorm eps 1e-3 / momentum 0.01, attribute names (``_conv_stem``, ``_bn0``, ``_blocks.{i}._expand_conv`` ...) as upstream
# so that state_dict keys equal a reference checkpoint's; ``_conv_head`` / ``_bn1`` stay in the module (smp deletes only ``_fc``) and never
# run.  Pinned by the published parameter counts (tests/test_oracle.py).
_EFFNET_BLOCKS = [   # (repeats, kernel, stride, expand, in, out): efficientnet_pytorch's BlockDecoder strings r1_k3_s11_e1_i32_o16_se0.25 ...
    (1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320)]
_EFFNET_CFG = {   # width, depth, nominal image size; smp's stage_idxs
    'efficientnet-b0': dict(w=1.0, d=1.0, size=224, stage_idxs=(3, 5, 9, 16)),
    'efficientnet-b5': dict(w=1.6, d=2.2, size=456, stage_idxs=(8, 13, 27, 39)),
    'efficientnet-b7': dict(w=2.0, d=3.1, size=600, stage_idxs=(11, 18, 38, 55)),
}


def effnet_round_filters(filters, w, divisor=8):
    filters *= w
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def effnet_block_list(name):
    """[(kernel, stride, expand, cin, cout, se_channels, (pad_top, pad_bottom) of the depthwise conv)] per block, plus the stem / head widths
    and the stem's padding: what EfficientNet.__init__ builds, static paddings included."""
    cfg = _EFFNET_CFG[name]
    size = cfg['size']

    def same_pad(ih, k, s):
        oh = -(-ih // s)
        pad = max((oh - 1) * s + (k - 1) + 1 - ih, 0)
        return (pad // 2, pad - pad // 2)

    stem_pad = same_pad(size, 3, 2)
    size = -(-size // 2)
    blocks = []
    for rep, k, st, e, cin, cout in _EFFNET_BLOCKS:
        cin, cout = effnet_round_filters(cin, cfg['w']), effnet_round_filters(cout, cfg['w'])
        rep = int(math.ceil(cfg['d'] * rep))
        for j in range(rep):
            s_j = st if j == 0 else 1
            c_j = cin if j == 0 else cout
            blocks.append((k, s_j, e, c_j, cout, max(1, int(c_j * 0.25)), same_pad(size, k, s_j)))
            size = -(-size // s_j)
    return dict(stem=effnet_round_filters(32, cfg['w']), stem_pad=stem_pad, head=effnet_round_filters(1280, cfg['w']), blocks=blocks)


for _n, _c in _EFFNET_CFG.items():
    _bl = effnet_block_list(_n)
    ENCODER_CHANNELS[_n] = (3, _bl['stem']) + tuple(_bl['blocks'][i - 1][4] for i in _c['stage_idxs'])


class StaticSamePadConv2d(nn.Conv2d):
    """efficientnet_pytorch Conv2dStaticSamePadding: ZeroPad2d fixed at construction (pad = (top, bottom), the same for columns), then the
    conv without padding."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False, pad=(0, 0)):
        super().__init__(cin, cout, k, stride, 0, groups=groups, bias=bias)
        self.pad = pad

    def forward(self, x):
        if self.pad != (0, 0):
            x = F.pad(x, (self.pad[0], self.pad[1], self.pad[0], self.pad[1]))
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


class MBConvBlock(nn.Module):
    def __init__(self, k, stride, expand, cin, cout, se_ch, pad):
        super().__init__()
        mid = cin * expand
        self.expand = expand != 1
        self.id_skip = stride == 1 and cin == cout
        bn = dict(momentum=0.01, eps=1e-3)
        if self.expand:
            self._expand_conv = StaticSamePadConv2d(cin, mid, 1)
            self._bn0 = nn.BatchNorm2d(mid, **bn)
        self._depthwise_conv = StaticSamePadConv2d(mid, mid, k, stride, groups=mid, pad=pad)
        self._bn1 = nn.BatchNorm2d(mid, **bn)
        self._se_reduce = StaticSamePadConv2d(mid, se_ch, 1, bias=True)
        self._se_expand = StaticSamePadConv2d(se_ch, mid, 1, bias=True)
        self._project_conv = StaticSamePadConv2d(mid, cout, 1)
        self._bn2 = nn.BatchNorm2d(cout, **bn)
        self.drop_mask = None          # [B] float 0 / 1: injected drop_connect pattern of the next training forward (None: draw)

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand:
            x = F.silu(self._bn0(self._expand_conv(x)))
        x = F.silu(self._bn1(self._depthwise_conv(x)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(F.silu(self._se_reduce(sq)))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        if self.id_skip:
            if drop_connect_rate and self.training:
                keep = 1.0 - drop_connect_rate
                m = self.drop_mask if self.drop_mask is not None else torch.floor(keep + torch.rand(x.shape[0]))
                x = x / keep * m.to(x.dtype).view(-1, 1, 1, 1)
            x = x + inputs
        return x


class EfficientNetEncoder(nn.Module):
    """smp EfficientNetEncoder.get_stages / forward: [identity, stem + bn0 + swish, blocks[:i0], [i0:i1], [i1:i2], [i2:]], the drop_connect
    rate of block b = 0.2 * b / len(blocks)."""

    def __init__(self, name, in_channels=3, depth=5):
        super().__init__()
        cfg, bl = _EFFNET_CFG[name], effnet_block_list(name)
        self._depth = depth
        self._stage_idxs = cfg['stage_idxs']
        self.out_channels = ENCODER_CHANNELS[name][:depth + 1]
        bn = dict(momentum=0.01, eps=1e-3)
        self._conv_stem = StaticSamePadConv2d(in_channels, bl['stem'], 3, 2, pad=bl['stem_pad'])
        self._bn0 = nn.BatchNorm2d(bl['stem'], **bn)
        self._blocks = nn.ModuleList([MBConvBlock(*b) for b in bl['blocks']])
        self._conv_head = StaticSamePadConv2d(bl['blocks'][-1][4], bl['head'], 1)     # (kept by smp, never run)
        self._bn1 = nn.BatchNorm2d(bl['head'], **bn)
        self.drop_connect_rate = 0.2

    def forward(self, x):
        feats = [x]
        x = F.silu(self._bn0(self._conv_stem(x))); feats.append(x)
        lo, n = 0, float(len(self._blocks))
        for i in range(2, self._depth + 1):
            hi = self._stage_idxs[i - 2] if i - 2 < 3 else len(self._blocks)
            for b in range(lo, hi):
                x = self._blocks[b](x, self.drop_connect_rate * b / n)
            lo = hi
            feats.append(x)
        return feats


class Conv2dReLU(nn.Sequential):
    def __init__(self, cin, cout, kernel_size, padding=0):
        super().__init__(
            nn.Conv2d(cin, cout, kernel_size, padding=padding, bias=False),
            nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True),
        )


class UnetDecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = Conv2dReLU(cin + cskip, cout, 3, 1)
        self.attention1 = nn.Identity()
        self.conv2 = Conv2dReLU(cout, cout, 3, 1)
        self.attention2 = nn.Identity()

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=(256, 128, 64, 32, 16)):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        ins = [enc[0]] + list(decoder_channels[:-1])
        skips = enc[1:] + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList(
            [UnetDecoderBlock(i, s, o) for i, s, o in zip(ins, skips, decoder_channels)])

    def forward(self, *features):
        features = features[1:][::-1]
        x = self.center(features[0])
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UnetPlusPlusDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=(256, 128, 64, 32, 16)):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        self.in_channels = [enc[0]] + list(decoder_channels[:-1])
        self.skip_channels = enc[1:] + [0]
        self.out_channels = list(decoder_channels)
        blocks = {}
        for layer_idx in range(len(self.in_channels) - 1):
            for depth_idx in range(layer_idx + 1):
                if depth_idx == 0:
                    in_ch = self.in_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1)
                    out_ch = self.out_channels[layer_idx]
                else:
                    out_ch = self.skip_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1 - depth_idx)
                    in_ch = self.skip_channels[layer_idx - 1]
                blocks[f'x_{depth_idx}_{layer_idx}'] = UnetDecoderBlock(in_ch, skip_ch, out_ch)
        blocks[f'x_0_{len(self.in_channels) - 1}'] = UnetDecoderBlock(
            self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1

    def forward(self, *features):
        features = features[1:][::-1]
        dense = {}
        for layer_idx in range(len(self.in_channels) - 1):
            for depth_idx in range(self.depth - layer_idx):
                if layer_idx == 0:
                    dense[f'x_{depth_idx}_{depth_idx}'] = self.blocks[f'x_{depth_idx}_{depth_idx}'](
                        features[depth_idx], features[depth_idx + 1])
                else:
                    li = depth_idx + layer_idx
                    cat = [dense[f'x_{idx}_{li}'] for idx in range(depth_idx + 1, li + 1)]
                    cat = torch.cat(cat + [features[li + 1]], dim=1)
                    dense[f'x_{depth_idx}_{li}'] = self.blocks[f'x_{depth_idx}_{li}'](
                        dense[f'x_{depth_idx}_{li - 1}'], cat)
        dense[f'x_0_{self.depth}'] = self.blocks[f'x_0_{self.depth}'](dense[f'x_0_{self.depth - 1}'])
        return dense[f'x_0_{self.depth}']



# ------------------------------------------------------------------------------------------------ MAnet decoder
# smp 0.3.3 ``decoders/manet/decoder.py`` (reference sweep: configs/tune.yaml:17 ``MAnet`` through smp.create_model, model.py:38-44) at its
# defaults: decoder_channels (256, 128, 64, 32, 16), decoder_pab_channels 64, reduction 16, BatchNorm on.  Restated from the published source,
# its two quirks included: PAB's softmax runs over ALL (HW)^2 entries of the position map at once (``view(bsize, -1)`` + Softmax(dim=1)), and
# the attended map [B, HW, C] is ``reshape``d to [B, C, h, w] WITHOUT a transpose before it is added to the input.
class PAB(nn.Module):
    def __init__(self, in_channels, out_channels, pab_channels=64):
        super().__init__()
        self.pab_channels, self.in_channels = pab_channels, in_channels
        self.top_conv = nn.Conv2d(in_channels, pab_channels, kernel_size=1)
        self.center_conv = nn.Conv2d(in_channels, pab_channels, kernel_size=1)
        self.bottom_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1)
        self.map_softmax = nn.Softmax(dim=1)
        self.out_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1)

    def forward(self, x):
        bsize, h, w = x.size(0), x.size(2), x.size(3)
        x_top = self.top_conv(x).flatten(2)
        x_center = self.center_conv(x).flatten(2).transpose(1, 2)
        x_bottom = self.bottom_conv(x).flatten(2).transpose(1, 2)
        sp_map = torch.matmul(x_center, x_top)
        sp_map = self.map_softmax(sp_map.view(bsize, -1)).view(bsize, h * w, h * w)
        sp_map = torch.matmul(sp_map, x_bottom)
        sp_map = sp_map.reshape(bsize, self.in_channels, h, w)
        return self.out_conv(x + sp_map)


class MFAB(nn.Module):
    def __init__(self, in_channels, skip_channels, out_channels, reduction=16):
        super().__init__()
        self.hl_conv = nn.Sequential(Conv2dReLU(in_channels, in_channels, 3, 1), Conv2dReLU(in_channels, skip_channels, 1))
        rd = max(1, skip_channels // reduction)

        def se():
            return nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(skip_channels, rd, 1), nn.ReLU(inplace=True), nn.Conv2d(rd, skip_channels, 1), nn.Sigmoid())
        self.SE_ll = se()
        self.SE_hl = se()
        self.conv1 = Conv2dReLU(skip_channels + skip_channels, out_channels, 3, 1)
        self.conv2 = Conv2dReLU(out_channels, out_channels, 3, 1)

    def forward(self, x, skip=None):
        x = self.hl_conv(x)
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        attention_hl = self.SE_hl(x)
        if skip is not None:
            attention_hl = attention_hl + self.SE_ll(skip)
            x = x * attention_hl
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class MAnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=(256, 128, 64, 32, 16), reduction=16, pab_channels=64):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        head = enc[0]
        ins = [head] + list(decoder_channels[:-1])
        skips = enc[1:] + [0]
        self.center = PAB(head, head, pab_channels=pab_channels)
        self.blocks = nn.ModuleList([MFAB(i, s, o, reduction) if s > 0 else UnetDecoderBlock(i, s, o) for i, s, o in zip(ins, skips, decoder_channels)])

    def forward(self, *features):
        features = features[1:][::-1]
        x = self.center(features[0])
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x



# ------------------------------------------------------------------------------------------------ PAN decoder
# smp 0.3.3 ``decoders/pan`` (reference sweep: configs/tune.yaml:18 ``PAN`` through smp.create_model, model.py:38-44) at its defaults:
# encoder_output_stride 16 (the ResNet's layer4 dilated), decoder_channels 32, bilinear upsampling with align_corners=True, 3x3 head +
# UpsamplingBilinear2d(4).  Restated from the published source.
class PanConvBnRelu(nn.Module):
    def __init__(self, cin, cout, k, padding=0, add_relu=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, 1, padding, bias=True)
        self.bn = nn.BatchNorm2d(cout)
        self.add_relu = add_relu

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.add_relu else x


class FPABlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.branch1 = nn.Sequential(nn.AdaptiveAvgPool2d(1), PanConvBnRelu(cin, cout, 1))
        self.mid = nn.Sequential(PanConvBnRelu(cin, cout, 1))
        self.down1 = nn.Sequential(nn.MaxPool2d(2, 2), PanConvBnRelu(cin, 1, 7, 3))
        self.down2 = nn.Sequential(nn.MaxPool2d(2, 2), PanConvBnRelu(1, 1, 5, 2))
        self.down3 = nn.Sequential(nn.MaxPool2d(2, 2), PanConvBnRelu(1, 1, 3, 1), PanConvBnRelu(1, 1, 3, 1))
        self.conv2 = PanConvBnRelu(1, 1, 5, 2)
        self.conv1 = PanConvBnRelu(1, 1, 7, 3)

    def forward(self, x):
        h, w = x.size(2), x.size(3)
        up = dict(mode='bilinear', align_corners=True)
        b1 = F.interpolate(self.branch1(x), size=(h, w), **up)
        mid = self.mid(x)
        x1 = self.down1(x)
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x3 = F.interpolate(x3, size=(h // 4, w // 4), **up)
        x = self.conv2(x2) + x3
        x = F.interpolate(x, size=(h // 2, w // 2), **up)
        x = x + self.conv1(x1)
        x = F.interpolate(x, size=(h, w), **up)
        return torch.mul(x, mid) + b1


class GAUBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.AdaptiveAvgPool2d(1), PanConvBnRelu(cout, cout, 1, add_relu=False), nn.Sigmoid())
        self.conv2 = PanConvBnRelu(cin, cout, 3, 1)

    def forward(self, x, y):
        h, w = x.size(2), x.size(3)
        y_up = F.interpolate(y, size=(h, w), mode='bilinear', align_corners=True)
        return y_up + torch.mul(self.conv2(x), self.conv1(y))


class PANDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=32):
        super().__init__()
        self.out_channels = decoder_channels
        self.fpa = FPABlock(encoder_channels[-1], decoder_channels)
        self.gau3 = GAUBlock(encoder_channels[-2], decoder_channels)
        self.gau2 = GAUBlock(encoder_channels[-3], decoder_channels)
        self.gau1 = GAUBlock(encoder_channels[-4], decoder_channels)

    def forward(self, *features):
        x5 = self.fpa(features[-1])
        x4 = self.gau3(features[-2], x5)
        x3 = self.gau2(features[-3], x4)
        return self.gau1(features[-4], x3)


class TransposeX2(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(
            nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1),
            nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True),
        )


class LinknetDecoderBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Sequential(
            Conv2dReLU(cin, cin // 4, 1),
            TransposeX2(cin // 4, cin // 4),
            Conv2dReLU(cin // 4, cout, 1),
        )

    def forward(self, x, skip=None):
        x = self.block(x)
        if skip is not None:
            x = x + skip
        return x


class LinknetDecoder(nn.Module):
    def __init__(self, encoder_channels, prefinal_channels=32, n_blocks=5):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        ch = enc + [prefinal_channels]
        self.blocks = nn.ModuleList([LinknetDecoderBlock(ch[i], ch[i + 1]) for i in range(n_blocks)])

    def forward(self, *features):
        features = features[1:][::-1]
        x = features[0]
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


# ---- smp 0.3.3 decoders/fpn/decoder.py (sweep architecture `FPN`, reference configs/tune.yaml:9-18).  Restated from
# the published package (absent here): FPNBlock = nearest x2 of the coarser pyramid level + 1x1 skip conv (with bias);
# SegmentationBlock = Conv3x3GNReLU x max(1, n_upsamples), each 3x3 conv (no bias) + GroupNorm(32) + ReLU + bilinear x2
# (align_corners=True) while upsampling; MergeBlock('add'); Dropout2d(0.2); head = 1x1 conv + UpsamplingBilinear2d(4).
class Conv3x3GNReLU(nn.Module):
    def __init__(self, cin, cout, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.block = nn.Sequential(
            nn.Conv2d(cin, cout, (3, 3), stride=1, padding=1, bias=False),
            nn.GroupNorm(32, cout),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        x = self.block(x)
        if self.upsample:
            x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
        return x


class FPNBlock(nn.Module):
    def __init__(self, pyramid_channels, skip_channels):
        super().__init__()
        self.skip_conv = nn.Conv2d(skip_channels, pyramid_channels, kernel_size=1)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        return x + self.skip_conv(skip)


class FPNSegmentationBlock(nn.Module):
    def __init__(self, cin, cout, n_upsamples=0):
        super().__init__()
        blocks = [Conv3x3GNReLU(cin, cout, upsample=bool(n_upsamples))]
        for _ in range(1, n_upsamples):
            blocks.append(Conv3x3GNReLU(cout, cout, upsample=True))
        self.block = nn.Sequential(*blocks)

    def forward(self, x):
        return self.block(x)


class InjectableDropout2d(nn.Module):
    """nn.Dropout2d(p) whose keep pattern can be supplied: ``mask`` [N, C] of {0, 1} -- kept channels are scaled by 1 / (1 - p)
    exactly as torch does.  With no mask set it IS nn.functional.dropout2d (random in training, identity in eval)."""

    def __init__(self, p=0.2):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if not self.training:
            return x
        if self.mask is None:
            return F.dropout2d(x, self.p, True)
        return x * (self.mask.to(x.dtype) / (1.0 - self.p))[:, :, None, None]


class FPNDecoder(nn.Module):
    def __init__(self, encoder_channels, pyramid_channels=256, segmentation_channels=128, dropout=0.2):
        super().__init__()
        enc = list(encoder_channels)[::-1]
        self.out_channels = segmentation_channels
        self.p5 = nn.Conv2d(enc[0], pyramid_channels, kernel_size=1)
        self.p4 = FPNBlock(pyramid_channels, enc[1])
        self.p3 = FPNBlock(pyramid_channels, enc[2])
        self.p2 = FPNBlock(pyramid_channels, enc[3])
        self.seg_blocks = nn.ModuleList([FPNSegmentationBlock(pyramid_channels, segmentation_channels, n) for n in (3, 2, 1, 0)])
        self.dropout = InjectableDropout2d(dropout)

    def forward(self, *features):
        c2, c3, c4, c5 = features[-4:]
        p5 = self.p5(c5)
        p4 = self.p4(p5, c4)
        p3 = self.p3(p4, c3)
        p2 = self.p2(p3, c2)
        pyramid = [blk(p) for blk, p in zip(self.seg_blocks, (p5, p4, p3, p2))]
        return self.dropout(sum(pyramid))       # MergeBlock('add')


class InjectableDropout(nn.Module):
    """nn.Dropout(p) (element-wise) whose keep pattern can be supplied: ``mask`` of the input's shape, {0, 1}; kept elements are scaled by
    1 / (1 - p) as torch does.  With no mask set it IS nn.functional.dropout."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if not self.training:
            return x
        if self.mask is None:
            return F.dropout(x, self.p, True)
        return x * (self.mask.to(x.dtype) / (1.0 - self.p))


class SeparableConv2d(nn.Sequential):
    """smp base/modules.py SeparableConv2d: depthwise conv (groups = channels, no bias) then pointwise 1x1; nothing in between."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__(
            nn.Conv2d(cin, cin, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=cin, bias=False),
            nn.Conv2d(cin, cout, kernel_size=1, bias=bias),
        )


class ASPPSeparableConv(nn.Sequential):
    def __init__(self, cin, cout, dilation):
        super().__init__(SeparableConv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for mod in self:
            x = mod(x)
        return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


class ASPPConv(nn.Sequential):
    def __init__(self, cin, cout, dilation):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class ASPP(nn.Module):
    """smp decoders/deeplabv3/decoder.py ASPP: 1x1, three dilated 3x3 (separable for DeepLabV3+, dense for DeepLabV3), image pooling;
    concat; 1x1 project + BN + ReLU + Dropout(0.5) (element-wise)."""

    def __init__(self, cin, cout, atrous_rates, separable=True):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())]
        for r in atrous_rates:
            mods.append(ASPPSeparableConv(cin, cout, r) if separable else ASPPConv(cin, cout, r))
        mods.append(ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * cout, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(), InjectableDropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))


class DeepLabV3Decoder(nn.Sequential):
    """smp DeepLabV3Decoder: dense ASPP on the last feature (output stride 8), 3x3 conv + BN + ReLU."""

    def __init__(self, cin, out_channels=256, atrous_rates=(12, 24, 36)):
        super().__init__(ASPP(cin, out_channels, atrous_rates, separable=False),
                         nn.Conv2d(out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU())
        self.out_channels = out_channels

    @property
    def dropout(self):
        return self[0].project[3]

    def forward(self, *features):
        return super().forward(features[-1])


class DeepLabV3PlusDecoder(nn.Module):
    """smp decoders/deeplabv3/decoder.py DeepLabV3PlusDecoder (output_stride 16): ASPP on the stride-16 feature, separable 3x3, bilinear
    x4 (align_corners=True), concat with a 48-channel 1x1 projection of the stride-4 feature, separable 3x3."""

    def __init__(self, encoder_channels, out_channels=256, atrous_rates=(12, 24, 36), output_stride=16):
        super().__init__()
        if output_stride not in (8, 16):
            raise ValueError(f'Output stride should be 8 or 16, got {output_stride}.')
        self.out_channels = out_channels
        self.aspp = nn.Sequential(
            ASPP(encoder_channels[-1], out_channels, atrous_rates),
            SeparableConv2d(out_channels, out_channels, 3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(),
        )
        self.up = nn.UpsamplingBilinear2d(scale_factor=2 if output_stride == 8 else 4)
        hi_in, hi_out = encoder_channels[-4], 48
        self.block1 = nn.Sequential(nn.Conv2d(hi_in, hi_out, 1, bias=False), nn.BatchNorm2d(hi_out), nn.ReLU())
        self.block2 = nn.Sequential(SeparableConv2d(hi_out + out_channels, out_channels, 3, padding=1, bias=False),
                                    nn.BatchNorm2d(out_channels), nn.ReLU())

    @property
    def dropout(self):
        return self.aspp[0].project[3]

    def forward(self, *features):
        a = self.up(self.aspp(features[-1]))
        hi = self.block1(features[-4])
        return self.block2(torch.cat([a, hi], dim=1))


class PSPBlock(nn.Module):
    """smp decoders/pspnet/decoder.py PSPBlock: adaptive average pooling to pool_size^2 bins, 1x1 Conv2dReLU (BatchNorm unless pool_size is
    1 -- then a biased conv), bilinear resize back (align_corners=True)."""

    def __init__(self, cin, cout, pool_size, use_batchnorm=True):
        super().__init__()
        if pool_size == 1:
            use_batchnorm = False
        conv = nn.Conv2d(cin, cout, 1, bias=not use_batchnorm)
        self.pool = nn.Sequential(nn.AdaptiveAvgPool2d((pool_size, pool_size)),
                                  nn.Sequential(conv, nn.BatchNorm2d(cout) if use_batchnorm else nn.Identity(), nn.ReLU(inplace=True)))

    def forward(self, x):
        h, w = x.shape[-2:]
        return F.interpolate(self.pool(x), size=(h, w), mode='bilinear', align_corners=True)


class PSPModule(nn.Module):
    def __init__(self, cin, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.blocks = nn.ModuleList([PSPBlock(cin, cin // len(sizes), s) for s in sizes])

    def forward(self, x):
        return torch.cat([b(x) for b in self.blocks] + [x], dim=1)


class PSPDecoder(nn.Module):
    """smp PSPDecoder: pyramid pooling on the LAST encoder feature (encoder_depth 3: stride 8), 1x1 Conv2dReLU to 512, Dropout2d(0.2)."""

    def __init__(self, encoder_channels, out_channels=512, dropout=0.2):
        super().__init__()
        self.out_channels = out_channels
        self.psp = PSPModule(encoder_channels[-1])
        self.conv = nn.Sequential(nn.Conv2d(encoder_channels[-1] * 2, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))
        self.dropout = InjectableDropout2d(dropout)

    def forward(self, *features):
        return self.dropout(self.conv(self.psp(features[-1])))


class SegmentationHead(nn.Sequential):
    def __init__(self, cin, cout, kernel_size, upsampling=1):
        super().__init__(
            nn.Conv2d(cin, cout, kernel_size, padding=kernel_size // 2),
            nn.UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity(),
            nn.Identity(),
        )


def _init_decoder(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, mode='fan_in', nonlinearity='relu')
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _init_head(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class SegmentationModel(nn.Module):
    def __init__(self, arch, encoder_name, in_channels, classes):
        super().__init__()
        if encoder_name in _EFFNET_CFG:
            if arch in ('deeplabv3', 'deeplabv3plus'):
                raise ValueError('the dilated (make_dilated) EfficientNet encoders are not restated (smp raises for them as well)')
            self.encoder = EfficientNetEncoder(encoder_name, in_channels, depth=3 if arch == 'pspnet' else 5)
        elif encoder_name in _REGNET_CFG:
            if arch in ('deeplabv3', 'deeplabv3plus'):
                raise ValueError('the dilated (make_dilated) RegNet encoders are not restated')
            self.encoder = RegNetEncoder(encoder_name, in_channels, depth=3 if arch == 'pspnet' else 5)
        else:
            self.encoder = ResNetEncoder(encoder_name, in_channels, depth=3 if arch == 'pspnet' else 5)
        ch = self.encoder.out_channels
        if arch == 'unet':
            self.decoder = UnetDecoder(ch)
            self.segmentation_head = SegmentationHead(16, classes, 3)
        elif arch == 'unetplusplus':
            self.decoder = UnetPlusPlusDecoder(ch)
            self.segmentation_head = SegmentationHead(16, classes, 3)
        elif arch == 'pan':
            if not isinstance(self.encoder, ResNetEncoder):
                raise ValueError('PAN dilates its encoder (output stride 16): restated for the ResNets')
            self.encoder.make_dilated(16)
            self.decoder = PANDecoder(ch)
            self.segmentation_head = SegmentationHead(self.decoder.out_channels, classes, 3, upsampling=4)
        elif arch == 'manet':
            self.decoder = MAnetDecoder(ch)
            self.segmentation_head = SegmentationHead(16, classes, 3)
        elif arch == 'linknet':
            self.decoder = LinknetDecoder(ch)
            self.segmentation_head = SegmentationHead(32, classes, 1)
        elif arch == 'fpn':
            self.decoder = FPNDecoder(ch)
            self.segmentation_head = SegmentationHead(self.decoder.out_channels, classes, 1, upsampling=4)
        elif arch == 'pspnet':
            # smp PSPNet defaults: encoder_depth=3, psp_out_channels=512, psp_use_batchnorm=True, psp_dropout=0.2, upsampling=8, 3x3 head
            self.decoder = PSPDecoder(ch)
            self.segmentation_head = SegmentationHead(self.decoder.out_channels, classes, 3, upsampling=8)
        elif arch == 'deeplabv3':
            # smp DeepLabV3 defaults: output stride 8 (layer3 dilation 2, layer4 dilation 4), decoder_channels=256, upsampling=8
            self.encoder.make_dilated(8)
            self.decoder = DeepLabV3Decoder(ch[-1])
            self.segmentation_head = SegmentationHead(self.decoder.out_channels, classes, 1, upsampling=8)
        elif arch == 'deeplabv3plus':
            # smp DeepLabV3Plus defaults: encoder_output_stride=16, decoder_channels=256, decoder_atrous_rates=(12, 24, 36), upsampling=4
            self.encoder.make_dilated(16)
            self.decoder = DeepLabV3PlusDecoder(ch)
            self.segmentation_head = SegmentationHead(self.decoder.out_channels, classes, 1, upsampling=4)
        else:
            raise KeyError(arch)
        _init_decoder(self.decoder)
        _init_head(self.segmentation_head)

    def forward(self, x):
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            raise RuntimeError(
                f'Wrong input shape height={h}, width={w}. Expected image height and width '
                f'divisible by 32.')
        return self.segmentation_head(self.decoder(*self.encoder(x)))


_ARCHS = ('unet', 'unetplusplus', 'linknet', 'fpn', 'deeplabv3plus', 'pspnet', 'deeplabv3', 'manet', 'pan')


def create_model(arch, encoder_name='resnet34', encoder_weights=None, in_channels=3, classes=1, **kwargs):
    """smp.create_model restated (reference src/models/smp/model.py:38-44).

    ``encoder_weights`` is accepted and ignored: no pretrained blobs exist
    offline, weights come from the seeded torch RNG.
    """
    a = arch.lower()
    if a not in _ARCHS:
        raise KeyError(f'Wrong architecture type `{arch}`. Available options are: {list(_ARCHS)}')
    if encoder_name not in ENCODER_CHANNELS:
        raise KeyError(f'Wrong encoder name `{encoder_name}`, supported: {list(ENCODER_CHANNELS)}')
    return SegmentationModel(a, encoder_name, in_channels, classes)


def conv_param_count(model):
    return sum(p.numel() for n, p in model.named_parameters() if p.dim() == 4)


def randomize_bn(model, seed=0):
    """Give BN affine params / running stats non-trivial values so parity tests
    exercise gamma/beta/running buffers (fresh init is 1/0/0/1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=g))
    return model


_ = math
