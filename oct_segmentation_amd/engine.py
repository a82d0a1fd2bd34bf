"""Plan-level host binding: ``SegNet`` = the MI355X counterpart of
``smp.create_model(arch, encoder_name, in_channels, classes)`` (reference
``src/models/smp/model.py:38-44``).

PyTorch is used for device memory only: one flat fp32 parameter arena (+ a
gradient arena of the same layout), a BN buffer arena and one workspace blob
per (B, H, W) plan.  All arithmetic happens in ``liboctseg_hip.so``.
"""
import ctypes as C
import os
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L

_ARCHS = ('unet', 'unetplusplus', 'linknet', 'fpn', 'deeplabv3plus', 'pspnet', 'deeplabv3', 'manet', 'pan')
_FPN_SEG_CHANNELS, _FPN_DROPOUT = 128, 0.2   # smp FPN defaults: decoder_segmentation_channels, decoder_dropout
_DLV3P_CHANNELS, _DLV3P_DROPOUT = 256, 0.5  # smp DeepLabV3Plus: decoder_channels, the nn.Dropout(0.5) of ASPP.project (element-wise)
_PSP_CHANNELS, _PSP_DROPOUT = 512, 0.2      # smp PSPNet: psp_out_channels, psp_dropout (Dropout2d)
_ENCODERS = ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152',
             # timm RegNet through smp's RegNetEncoder (reference configs/tune.yaml:19-24); grouped 3x3 convs run as per-group launches
             'timm-regnetx_002', 'timm-regnetx_064', 'timm-regnety_120',   # (RegNetY: + squeeze-excite gates, csrc/se.hip)
             # efficientnet_pytorch through smp's EfficientNetEncoder (configs/tune.yaml:25-28): MBConv blocks, csrc/effnet.hip
             'efficientnet-b0', 'efficientnet-b5', 'efficientnet-b7')
_EFFNET_STAGE1 = {'efficientnet-b0': 5, 'efficientnet-b5': 13, 'efficientnet-b7': 18}   # smp _stage_idxs[1]: first block behind the stride-8 feature


def get_preprocessing_params(encoder_name, pretrained='imagenet'):
    """smp.encoders.get_preprocessing_params (model.py:49): the torchvision ResNets and smp's timm-regnet* entries share the ImageNet
    statistics below."""
    if encoder_name not in _ENCODERS:
        raise KeyError(f'Wrong encoder name `{encoder_name}`, supported encoders: {list(_ENCODERS)}')
    return {'input_space': 'RGB', 'input_range': [0, 1], 'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}


def _dtype_code(dtype):
    if dtype in (torch.float32, 'fp32', 'f32', 'float32'):
        return L.F32
    if dtype in (torch.bfloat16, 'bf16', 'bfloat16'):
        return L.BF16
    if dtype in (torch.float16, 'fp16', 'f16', 'float16', 'half'):
        return L.F16   # serving only (eval forwards): BASELINE config #5
    raise TypeError(f'compute dtype must be float32, bfloat16 or float16, got {dtype}')


class _Plan:
    """RAII wrapper of an ``octseg_plan`` plus its workspace."""

    def __init__(self, arch, encoder, classes, B, H, W, dtype_code):
        self.handle = C.c_void_p()
        d = L.NetDesc(arch.encode(), encoder.encode(), classes, B, H, W, dtype_code)
        rc = L.lib().octseg_plan_create(C.byref(d), C.byref(self.handle))
        if rc != 0:
            msg = L.lib().octseg_last_error().decode()
            if rc == -3:
                raise KeyError(msg)
            raise RuntimeError(msg)  # -1: same text as smp's check_input_shape RuntimeError
        self.workspace = None
        self.shape = (B, H, W)
        self.generation = 0   # forwards run on this plan's workspace (a backward must follow ITS forward)

    def ws(self, device):
        if self.workspace is None:
            n = L.lib().octseg_plan_workspace_bytes(self.handle)
            self.workspace = torch.empty(n, dtype=torch.uint8, device=device)
        return self.workspace

    def __del__(self):
        try:
            if self.handle:
                L.lib().octseg_plan_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


class _DiceStep(torch.autograd.Function):
    """Connects the engine's fused forward+Dice to autograd so ``loss.backward()`` (what
    Lightning does after ``training_step``) fills the parameter arena's ``.grad``."""

    @staticmethod
    def forward(ctx, arena, net, image, target, normalize, mean, std):
        logits, loss, stats, plan = net._forward_loss(image, target, normalize, mean, std)
        ctx.net, ctx.plan, ctx.logits, ctx.target, ctx.generation = net, plan, logits, target, plan.generation
        # the stem's weight gradient gathers the frame again in the backward: keep alive the CONTIGUOUS tensor the kernel read
        # (for a channels_last / permuted input that is a copy made by _check_input, not `image` itself)
        ctx.image = plan.stem_frame
        ctx.mark_non_differentiable(logits, stats)
        return loss, logits, stats

    @staticmethod
    def backward(ctx, gloss, _gl, _gs):
        net = ctx.net
        g = net._backward(ctx.plan, ctx.logits, ctx.target, generation=ctx.generation)
        return g * gloss, None, None, None, None, None, None


class _LayerRef:
    """One module position of the reference's tree (``model.model.encoder.layer4[-1]``, the Grad-CAM target of
    ``src/models/visualize_activation_maps.py:103``).  The engine has no per-layer ``nn.Module`` objects -- the whole graph runs
    behind ``octseg_net_forward`` -- so this is a named handle: it knows its ``state_dict`` prefix and its parameter views, and it
    refuses forward hooks loudly instead of letting a CAM tool record nothing."""

    def __init__(self, net, name):
        self._net, self.name = net, name

    def named_parameters(self):
        pre = self.name + '.'
        return [(p['name'][len(pre):], self._net._torch_view(p)) for p in self._net.param_table if p['name'].startswith(pre)]

    def register_forward_hook(self, *_a, **_k):
        raise NotImplementedError(f'{self.name}: the gfx950 engine exposes no per-layer activations to hooks (Grad-CAM tooling is '
                                  f'outside the accelerated path); use engine.debug_tensor() for a conv output')
    register_full_backward_hook = register_forward_hook

    def __repr__(self):
        return f'<octseg layer {self.name}>'


class _TreeRef:
    """Attribute / index navigation over the parameter names: ``.encoder.layer4[-1].conv3``."""

    def __init__(self, net, prefix):
        self._net, self._prefix = net, prefix

    def _children(self):
        pre = self._prefix + '.'
        return sorted({p['name'][len(pre):].split('.')[0] for p in self._net.param_table if p['name'].startswith(pre)},
                      key=lambda k: (0, int(k)) if k.isdigit() else (1, k))

    def __getattr__(self, key):
        if key.startswith('_'):
            raise AttributeError(key)
        if key not in self._children():
            raise AttributeError(f'{self._prefix} has no submodule {key!r} (has: {self._children()})')
        return _TreeRef(self._net, f'{self._prefix}.{key}')

    def __len__(self):
        return len([k for k in self._children() if k.isdigit()])

    def __getitem__(self, i):
        idx = [k for k in self._children() if k.isdigit()]
        return _TreeRef(self._net, f'{self._prefix}.{idx[i]}')

    def ref(self):
        return _LayerRef(self._net, self._prefix)

    def register_forward_hook(self, *a, **k):
        return self.ref().register_forward_hook(*a, **k)

    def __repr__(self):
        return f'<octseg module path {self._prefix}: {self._children()}>'


class SegNet(nn.Module):
    """Segmentation network living in ``liboctseg_hip.so``.

    ``forward(x)`` takes NCHW float32 ``[B, 3, H, W]`` on the GPU and returns NCHW float32
    logits ``[B, classes, H, W]``.  ``state_dict()`` / ``load_state_dict()`` speak the smp /
    torchvision key names and torch weight layouts of a reference checkpoint.
    """

    # smp.create_model keywords this engine implements at their smp 0.3.3 defaults only (anything else changes the graph)
    _SMP_DEFAULTS = {'encoder_depth': 5, 'decoder_use_batchnorm': True, 'decoder_channels': (256, 128, 64, 32, 16),
                     'decoder_attention_type': None, 'activation': None, 'aux_params': None,
                     # smp.FPN's own keywords, at their defaults
                     'decoder_pyramid_channels': 256, 'decoder_segmentation_channels': 128, 'decoder_merge_policy': 'add',
                     'decoder_dropout': 0.2, 'upsampling': 4,
                     # smp.DeepLabV3Plus's own keywords, at their defaults (decoder_channels: see _ARCH_DEFAULTS)
                     'encoder_output_stride': 16, 'decoder_atrous_rates': (12, 24, 36),
                     # smp.PSPNet's own keywords, at their defaults (encoder_depth, upsampling: see _ARCH_DEFAULTS)
                     'psp_out_channels': 512, 'psp_use_batchnorm': True, 'psp_dropout': 0.2,
                     # smp.MAnet's own keywords, at their defaults
                     'decoder_pab_channels': 64}
    _ARCH_DEFAULTS = {'deeplabv3plus': {'decoder_channels': 256}, 'pspnet': {'encoder_depth': 3, 'upsampling': 8}, 'pan': {'decoder_channels': 32},
                      'deeplabv3': {'decoder_channels': 256, 'upsampling': 8}}

    def __init__(self, arch, encoder_name='resnet34', encoder_weights=None, in_channels=3, classes=1,
                 device='cuda', compute_dtype=torch.bfloat16, seed=None, **kwargs):
        super().__init__()
        use_graph = bool(kwargs.pop('use_graph', False))   # eval forwards as replayed hipGraphs (serving)
        use_train_graph = bool(kwargs.pop('use_train_graph', False))
        loss = kwargs.pop('loss', 'dice')   # 'dice' (the reference, model.py:55) | 'bce' | 'dice+bce': octseg_plan_set_loss
        if loss not in L.LOSS_KINDS:
            raise ValueError(f'loss must be one of {list(L.LOSS_KINDS)}, got {loss!r}')
        for k, v in kwargs.items():
            if k not in self._SMP_DEFAULTS:
                raise TypeError(f'SegNet got an unexpected keyword argument {k!r}')
            dflt = self._ARCH_DEFAULTS.get(arch.lower(), {}).get(k, self._SMP_DEFAULTS[k])
            if (tuple(v) if isinstance(v, (list, tuple)) else v) != dflt:
                raise NotImplementedError(f'{k}={v!r}: the gfx950 engine builds the smp default ({dflt!r}) only')
        a = arch.lower()
        if a not in _ARCHS:
            raise KeyError(f'Wrong architecture type `{arch}`. Available options are: {list(_ARCHS)}')
        if encoder_name not in _ENCODERS:
            raise KeyError(f'Wrong encoder name `{encoder_name}`, supported encoders: {list(_ENCODERS)}')
        if in_channels != 3:
            raise ValueError('the gfx950 stem kernel is specialised for in_channels=3 (the reference always uses 3)')
        L.lib()  # fail loudly right here when the HIP library is missing
        self.arch, self.encoder_name, self.classes = a, encoder_name, int(classes)
        self.dtype_code = _dtype_code(compute_dtype)
        self.device = torch.device(device)
        self._plans = {}
        self.use_graph = use_graph
        self.loss = loss
        # train_step_raw as ONE replayed hipGraph per (B, H, W) plan (octseg_net_train_step + octseg_plan_set_train_graph): the ~800
        # launches of a step cost the host tens of milliseconds to enqueue, which bounds small per-GPU batches; not with `exchange`
        self.use_train_graph = use_train_graph
        # arch 'fpn': nn.Dropout2d(0.2) sits behind the merge; arch 'deeplabv3plus': nn.Dropout(0.5) behind ASPP.project.  Training
        # forwards draw a keep pattern on the device (torch's RNG, as the reference's modules do) -- [B, 128] per channel for FPN
        # ([B, 512] for PSPNet's Dropout2d(0.2)),
        # [B, H/16, W/16, 256] NHWC per element for DeepLabV3+ -- unless `dropout_keep` holds one (tests inject the oracle's;
        # DeepLabV3+ also accepts torch's NCHW [B, 256, H/16, W/16]); eval ignores it.
        self.dropout_keep = None
        # EfficientNet: drop_connect on the id skips of the MBConv blocks.  Training forwards draw floor(1 - rate + U[0, 1)) per block and
        # sample (torch's RNG) unless `drop_connect_keep` holds a [blocks][B] pattern of 0 / 1 (tests inject the oracle's); eval ignores it
        self.drop_connect_keep = None
        self._param_epoch = 0   # bumped by writers that bypass torch's version counter (the fused optimizer)
        self._buffer_epoch = 0  # the same for bn_buffers (every train-mode forward)
        # parameter table from a shape-independent probe plan (32x32 is the smallest legal input)
        probe = _Plan(a, encoder_name, self.classes, 1, 32, 32, self.dtype_code)
        lib = L.lib()
        self.param_table = []
        for i in range(lib.octseg_plan_num_params(probe.handle)):
            pi = L.ParamInfo()
            L.check(lib.octseg_plan_param_info(probe.handle, i, C.byref(pi)))
            self.param_table.append(dict(name=pi.name.decode(), kind=pi.kind, R=pi.R, S=pi.S, O=pi.O, I=pi.I, KP=pi.KP,
                                         offset=pi.offset, numel=pi.numel))
        self.bn_table = []
        for i in range(lib.octseg_plan_num_bn(probe.handle)):
            bi = L.BNInfo()
            L.check(lib.octseg_plan_bn_info(probe.handle, i, C.byref(bi)))
            self.bn_table.append(dict(name=bi.name.decode(), C=bi.C, mean_offset=bi.mean_offset, var_offset=bi.var_offset))
        self.param_numel = lib.octseg_plan_param_numel(probe.handle)
        self.buffer_numel = lib.octseg_plan_buffer_numel(probe.handle)
        del probe
        self.arena = nn.Parameter(torch.zeros(self.param_numel, dtype=torch.float32, device=self.device))
        self.register_buffer('bn_buffers', torch.zeros(self.buffer_numel, dtype=torch.float32, device=self.device))
        self.register_buffer('num_batches_tracked', torch.zeros((), dtype=torch.long))
        self._grad_arena = torch.zeros(self.param_numel, dtype=torch.float32, device=self.device)
        self._by_name = {p['name']: p for p in self.param_table}
        # grouped convs (timm RegNet conv2): the engine holds one parameter per group, named <key>#g<k>; torch holds ONE tensor
        # [Cout][gw][k][k] = the groups joined along dim 0.  _keyed: state_dict key -> the engine parameters behind it, in order
        self._keyed = OrderedDict()
        for p in self.param_table:
            self._keyed.setdefault(p['name'].split('#g')[0], []).append(p)
        # element ranges of the arena that an optimizer may touch.  PSPNet (smp encoder_depth 3) keeps encoder.layer3 / layer4 in its
        # state_dict without ever running them: torch optimizers skip parameters whose gradient is None, so weight decay must not
        # reach them here either -- the fused optimizer steps over the live ranges only
        dead = (('encoder.s3.', 'encoder.s4.') if encoder_name.startswith('timm-regnet') else ('encoder.layer3.', 'encoder.layer4.')) if a == 'pspnet' else ()
        if encoder_name.startswith('efficientnet-'):      # smp deletes only _fc: the classifier's conv / BatchNorm stay in state_dict and never run
            nblk = 1 + max(int(p['name'].split('.')[2]) for p in self.param_table if p['name'].startswith('encoder._blocks.'))
            dead = ('encoder._conv_head.', 'encoder._bn1.')
            if a == 'pspnet':
                dead += tuple(f'encoder._blocks.{k}.' for k in range(_EFFNET_STAGE1[encoder_name], nblk))
        self._dead_prefixes = dead
        self.live_ranges, lo = [], 0
        for p in sorted(self.param_table, key=lambda q: q['offset']):
            if p['name'].startswith(dead) if dead else False:
                if p['offset'] > lo:
                    self.live_ranges.append((lo, p['offset']))
                lo = p['offset'] + (p['numel'] + 3) // 4 * 4
        if lo < self.param_numel:
            self.live_ranges.append((lo, self.param_numel))
        self.initialize(seed)
        self.load_encoder_weights(encoder_weights)

    # the reference's attribute paths (``model.model.encoder.layer4[-1]``): name handles, see _TreeRef
    @property
    def encoder(self):
        return _TreeRef(self, 'encoder')

    @property
    def decoder(self):
        return _TreeRef(self, 'decoder')

    @property
    def segmentation_head(self):
        return _TreeRef(self, 'segmentation_head')

    def load_encoder_weights(self, encoder_weights):
        """``encoder_weights`` of smp.create_model.  The reference never passes it, so smp's default ``'imagenet'`` downloads
        torchvision's ResNet weights (model.py:38-44; train.py:17 patches ssl for that download).  This engine has no network
        path: ``None`` = random init (as ``predict.py:41``), a file path or a state_dict = torchvision ResNet weights
        (keys ``conv1.weight``, ``layer1.0.bn1.running_mean`` ..., ``fc.*`` ignored) loaded into ``encoder.*``;
        ``'imagenet'`` resolves to ``$OCTSEG_IMAGENET_DIR/<encoder_name>.pth`` and raises when that file is missing rather
        than silently training from scratch."""
        if encoder_weights is None:
            return
        if isinstance(encoder_weights, str) and not os.path.exists(encoder_weights):
            root = os.environ.get('OCTSEG_IMAGENET_DIR')
            cand = os.path.join(root, f'{self.encoder_name}.pth') if root else None
            if cand is None or not os.path.exists(cand):
                raise RuntimeError(
                    f"encoder_weights={encoder_weights!r}: no pretrained weights can be downloaded here. Pass the path (or the "
                    f"state_dict) of torchvision's {self.encoder_name} weights, or put {self.encoder_name}.pth into "
                    f"$OCTSEG_IMAGENET_DIR; encoder_weights=None starts from random init.")
            encoder_weights = cand
        sd = torch.load(encoder_weights, map_location='cpu', weights_only=True) if isinstance(encoder_weights, str) else encoder_weights
        sd = {('encoder.' + k): v for k, v in sd.items() if not k.startswith('fc.')}
        own = {k for k in self.state_dict().keys() if k.startswith('encoder.')}
        missing, unexpected = sorted(own - set(sd)), sorted(set(sd) - own)
        if missing or unexpected:
            raise RuntimeError(f'encoder weights do not fit {self.encoder_name}: missing {missing[:4]}, unexpected {unexpected[:4]}')
        self.load_state_dict(sd, strict=False)

    def _apply(self, fn, recurse=True):
        """``.to()`` / ``.cuda()``: the arenas move with the module, so the private gradient arena, the device the plans
        allocate on and the plans' workspaces follow; a dtype change is refused (the arenas are fp32 master copies)."""
        super()._apply(fn, recurse)
        if self.arena.dtype != torch.float32:
            raise TypeError('SegNet keeps fp32 master parameters; choose the compute dtype with compute_dtype=')
        if self.arena.device != self.device:
            if self.arena.device.type != 'cuda':
                raise RuntimeError('SegNet lives on the GPU (there is no CPU path)')
            self.device = self.arena.device
            self._grad_arena = self._grad_arena.to(self.device)
            self._plans = {}
            self.params_changed()
        return self

    # ------------------------------------------------------------------ parameter views
    def _torch_view(self, p, arena=None):
        """View of one parameter in torch's layout (no copy)."""
        arena = self.arena.data if arena is None else arena
        flat = arena[p['offset']:p['offset'] + p['numel']]
        if p['kind'] == L.P_CONV:
            return flat.view(p['R'], p['S'], p['O'], p['I']).permute(2, 3, 0, 1)
        if p['kind'] == L.P_CONVT:
            return flat.view(p['R'], p['S'], p['O'], p['I']).permute(3, 2, 0, 1)
        if p['kind'] == L.P_STEM:      # [O][KP] im2col rows, k = (r * S + s) * 3 + ci (7x7: 147 of 160; timm RegNet's 3x3 stem: 27 of 32)
            return flat.view(p['O'], p['KP'])[:, :p['R'] * p['S'] * 3].view(p['O'], p['R'], p['S'], 3).permute(0, 3, 1, 2)
        return flat

    def torch_shape(self, p):
        return tuple(self._torch_view(p).shape)

    def initialize(self, seed=None):
        """smp/torchvision initialisers: encoder kaiming_normal(fan_out), decoder
        kaiming_uniform(fan_in), ConvTranspose2d torch default, head xavier_uniform, BN 1/0."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        with torch.no_grad():
            self.arena.zero_()
            for p in self.param_table:
                shape = self.torch_shape(p)
                name = p['name']
                t = torch.empty(shape)
                if p['kind'] == L.P_VEC:
                    if name.endswith('.bias'):
                        t.zero_()
                        if p['name'].endswith('.1.0.bias') and '.block.' in name:  # ConvTranspose2d bias default
                            fan_in = self._by_name[name[:-4] + 'weight']['O'] * 16
                            bound = 1.0 / fan_in ** 0.5
                            t.uniform_(-bound, bound, generator=g)
                    else:
                        t.fill_(1.0)
                        if self.encoder_name.startswith('timm-regnet') and name.endswith('.conv3.bn.weight'):
                            t.zero_()      # timm RegNet(zero_init_last=True): the block's last BatchNorm starts at gamma = 0
                elif name.startswith('encoder.') and self.encoder_name.startswith('efficientnet-'):
                    nn.init.kaiming_uniform_(t, a=5 ** 0.5, generator=g)     # efficientnet_pytorch: torch's Conv2d default
                elif name.startswith('encoder.'):
                    nn.init.kaiming_normal_(t, mode='fan_out', nonlinearity='relu', generator=g)
                elif name.startswith('segmentation_head.'):
                    nn.init.xavier_uniform_(t, generator=g)
                elif p['kind'] == L.P_CONVT:
                    nn.init.kaiming_uniform_(t, a=5 ** 0.5, generator=g)
                else:
                    nn.init.kaiming_uniform_(t, mode='fan_in', nonlinearity='relu', generator=g)
                self._torch_view(p).copy_(t.to(self.device))
            for b in self.bn_table:
                self.bn_buffers[b['mean_offset']:b['mean_offset'] + b['C']] = 0.0
                self.bn_buffers[b['var_offset']:b['var_offset'] + b['C']] = 1.0
            self.num_batches_tracked.zero_()
        self.params_changed()

    # ------------------------------------------------------------------ state_dict in reference key space
    def state_dict(self, destination=None, prefix='', keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        bn_by_name = {b['name']: b for b in self.bn_table}
        for key, parts in self._keyed.items():
            p = parts[-1]
            sd[prefix + key] = (self._torch_view(p).detach().clone().contiguous() if len(parts) == 1 else
                                torch.cat([self._torch_view(q).detach() for q in parts], dim=0).contiguous())
            if p['name'].endswith('.bias') and p['name'][:-5] in bn_by_name:
                b = bn_by_name[p['name'][:-5]]
                base = prefix + b['name']
                sd[base + '.running_mean'] = self.bn_buffers[b['mean_offset']:b['mean_offset'] + b['C']].clone()
                sd[base + '.running_var'] = self.bn_buffers[b['var_offset']:b['var_offset'] + b['C']].clone()
                dead_bn = bool(self._dead_prefixes) and (b['name'] + '.').startswith(self._dead_prefixes)     # never ran: torch's counter stays 0
                sd[base + '.num_batches_tracked'] = torch.zeros_like(self.num_batches_tracked) if dead_bn else self.num_batches_tracked.clone()
        return sd

    def load_state_dict(self, state_dict, strict=True):
        expected = set(self.state_dict().keys())
        missing = sorted(expected - set(state_dict))
        unexpected = sorted(set(state_dict) - expected)
        if strict and (missing or unexpected):
            raise RuntimeError(f'Error(s) in loading state_dict for SegNet: missing keys {missing[:5]}..., '
                               f'unexpected keys {unexpected[:5]}...')
        with torch.no_grad():
            for key, parts in self._keyed.items():
                if key in state_dict:
                    src = state_dict[key]
                    views = [self._torch_view(q) for q in parts]
                    want = (sum(v.shape[0] for v in views),) + tuple(views[0].shape[1:])
                    if tuple(src.shape) != want:
                        raise RuntimeError(f"size mismatch for {key}: {tuple(src.shape)} vs {want}")
                    o = 0
                    for v in views:      # (one view for ordinary parameters; the groups of a grouped conv are slices along dim 0)
                        v.copy_(src[o:o + v.shape[0]].to(self.device, torch.float32))
                        o += v.shape[0]
            for b in self.bn_table:
                for key, off in (('running_mean', b['mean_offset']), ('running_var', b['var_offset'])):
                    k = f"{b['name']}.{key}"
                    if k in state_dict:
                        self.bn_buffers[off:off + b['C']] = state_dict[k].to(self.device, torch.float32)
                k = f"{b['name']}.num_batches_tracked"
                if k in state_dict and not (self._dead_prefixes and (b['name'] + '.').startswith(self._dead_prefixes)):
                    self.num_batches_tracked.copy_(state_dict[k])
        self.params_changed()   # the views write through arena.data, which torch's version counter does not see
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    def params_changed(self):
        """Tell every plan that the parameter arena was modified outside torch's view (raw pointer writes)."""
        self._param_epoch += 1

    def named_grads(self):
        """Gradients of the last backward, per parameter, in torch layout (for parity tests)."""
        src = self.arena.grad if self.arena.grad is not None else self._grad_arena
        return OrderedDict((key, torch.cat([self._torch_view(q, src).detach() for q in parts], dim=0).clone().contiguous())
                           for key, parts in self._keyed.items())

    # ------------------------------------------------------------------ execution
    def _plan(self, B, H, W):
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = _Plan(self.arch, self.encoder_name, self.classes, B, H, W, self.dtype_code)
        plan = self._plans[key]
        if getattr(plan, 'loss', 'dice') != self.loss:     # the criterion is a property of the net; plans follow it
            L.check(L.lib().octseg_plan_set_loss(plan.handle, L.LOSS_KINDS[self.loss]))
            plan.loss = self.loss
        return plan

    def fwd_macs(self, B, H, W):
        return L.lib().octseg_plan_fwd_macs(self._plan(B, H, W).handle)

    def exec_macs(self, B, H, W):
        """(forward, data gradient, weight gradient) multiply-accumulates a training step executes (octseg_plan_exec_macs)."""
        import ctypes
        out = (ctypes.c_double * 3)()
        L.check(L.lib().octseg_plan_exec_macs(self._plan(B, H, W).handle, out))
        return tuple(out)

    def _check_input(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise ValueError(f'expected a float32 CUDA tensor [B,3,H,W], got {tuple(x.shape)} {x.dtype} {x.device}')
        return x.contiguous()

    def _has_dropout(self):
        return self.arch in ('fpn', 'deeplabv3plus', 'pspnet', 'deeplabv3')

    def _keep_shape(self, B, H, W):
        if self.arch == 'fpn':
            return (B, _FPN_SEG_CHANNELS), 1.0 - _FPN_DROPOUT
        if self.arch == 'pspnet':
            return (B, _PSP_CHANNELS), 1.0 - _PSP_DROPOUT
        s = 8 if self.arch == 'deeplabv3' else 16          # DeepLabV3: output stride 8
        return (B, H // s, W // s, _DLV3P_CHANNELS), 1.0 - _DLV3P_DROPOUT

    def _draw_keep(self, B, H, W, device):
        """The dropout keep pattern of one training forward, in the layout octseg_plan_set_dropout takes."""
        shape, pkeep = self._keep_shape(B, H, W)
        keep = self.dropout_keep
        if keep is None:
            return torch.bernoulli(torch.full(shape, pkeep, device=device))
        keep = keep.to(device, torch.float32)
        if self.arch in ('deeplabv3plus', 'deeplabv3') and keep.dim() == 4 and tuple(keep.shape) == (shape[0], shape[3], shape[1], shape[2]):
            keep = keep.permute(0, 2, 3, 1)      # torch's NCHW mask -> the engine's NHWC
        keep = keep.contiguous()
        if tuple(keep.shape) != shape:
            raise ValueError(f'dropout_keep must be {list(shape)} of 0 / 1, got {tuple(keep.shape)}')
        return keep

    def _drop_connect_factors(self, plan, B, device):
        """EfficientNet: keep / (1 - rate) per id-skip block and sample -- floor(1 - rate + U[0, 1)) drawn here (efficientnet_pytorch's drop_connect),
        or the injected `drop_connect_keep` pattern."""
        lib = L.lib()
        nb = lib.octseg_plan_num_drop_connect(plan.handle)
        rates = torch.tensor([lib.octseg_plan_drop_connect_rate(plan.handle, i) for i in range(nb)], dtype=torch.float32, device=device)
        keep = self.drop_connect_keep
        if keep is None:
            keep = torch.floor((1.0 - rates).view(-1, 1) + torch.rand(nb, B, device=device))
        keep = keep.to(device, torch.float32)
        if tuple(keep.shape) != (nb, B):
            raise ValueError(f'drop_connect_keep must be [{nb}, {B}] of 0 / 1, got {tuple(keep.shape)}')
        return (keep / (1.0 - rates).view(-1, 1)).contiguous()

    def _graph_train_step(self, image, target, normalize, mean, std, grad_scale):
        """forward + Dice + backward through octseg_net_train_step with the plan's training graph on: inputs are copied into persistent
        buffers (a replay needs every pointer unchanged), the step runs on a capturable stream, outputs come back as copies."""
        x = self._check_input(image)
        B, _, H, W = x.shape
        plan = self._plan(B, H, W)
        plan.generation += 1
        ver = (self.arena._version, self._param_epoch)
        if getattr(plan, 'seen_version', None) != ver:
            L.check(L.lib().octseg_plan_params_changed(plan.handle))
            plan.seen_version = ver
        target = target.contiguous()
        if tuple(target.shape) != (B, self.classes, H, W) or target.dtype != torch.float32:
            raise ValueError(f'mask must be float32 {(B, self.classes, H, W)}, got {target.dtype} {tuple(target.shape)}')
        io = getattr(plan, 'train_io', None)
        if io is None:
            dev = x.device
            io = plan.train_io = dict(
                img=torch.empty_like(x), mask=torch.empty_like(target),
                logits=torch.empty((B, self.classes, H, W), dtype=torch.float32, device=dev),
                loss=torch.empty((), dtype=torch.float32, device=dev),
                stats=torch.empty((B, self.classes, 4), dtype=torch.int64, device=dev),
                keep=torch.ones(self._keep_shape(B, H, W)[0] if self._has_dropout() else (1,), dtype=torch.float32, device=dev),
                stream=torch.cuda.Stream(device=dev))    # (the legacy default stream cannot be captured)
            L.check(L.lib().octseg_plan_set_train_graph(plan.handle, 1))
            if self._has_dropout():
                L.check(L.lib().octseg_plan_set_dropout(plan.handle, L.ptr(io['keep'])))
        io['img'].copy_(x)
        io['mask'].copy_(target)
        if self._has_dropout():
            io['keep'].copy_(self._draw_keep(B, H, W, x.device))
        if self.encoder_name.startswith('efficientnet-'):     # drop_connect factors in a persistent buffer (a replay needs every pointer unchanged)
            fac = self._drop_connect_factors(plan, B, x.device)
            if 'dc' not in io:
                io['dc'] = torch.empty_like(fac)
                L.check(L.lib().octseg_plan_set_drop_connect(plan.handle, L.ptr(io['dc'])))
            io['dc'].copy_(fac)
        m = (C.c_float * 3)(*([float(v) for v in mean] if normalize else [0, 0, 0]))
        s = (C.c_float * 3)(*([float(v) for v in std] if normalize else [1, 1, 1]))
        cur = torch.cuda.current_stream(x.device)
        io['stream'].wait_stream(cur)
        with torch.cuda.stream(io['stream']):
            L.check(L.lib().octseg_net_train_step(plan.handle, L.ptr(self.arena.data), L.ptr(self._grad_arena), L.ptr(self.bn_buffers),
                                                  L.ptr(plan.ws(x.device)), L.ptr(io['img']), L.ptr(io['mask']), L.ptr(io['logits']),
                                                  L.ptr(io['loss']), L.ptr(io['stats']), int(bool(normalize)), m, s, float(grad_scale),
                                                  L.stream_ptr()))
        cur.wait_stream(io['stream'])
        self.num_batches_tracked += 1
        self._buffer_epoch += 1
        return io['loss'].clone(), io['logits'].clone(), io['stats'].clone()

    def _run_forward(self, x, normalize, mean, std, train, defer_join=False):
        x = self._check_input(x)
        B, _, H, W = x.shape
        plan = self._plan(B, H, W)
        plan.generation += 1   # this forward overwrites the plan's saved activations / BN statistics / Dice sums
        # in-place writes to the arena (torch optimizers, copy_, all-reduce) bump its version counter; the fused
        # optimizer calls params_changed() itself.  A changed version invalidates every plan's weight images.
        ver = (self.arena._version, self._param_epoch)
        # eval weight images fold the running statistics in: they are stale too once ANY plan's train-mode forward has updated
        # bn_buffers through the raw pointer (torch's version counter does not see that write: _buffer_epoch does)
        bver = (self.bn_buffers._version, self._buffer_epoch)
        if getattr(plan, 'seen_version', None) != ver or (not train and getattr(plan, 'seen_buffers', None) != bver):
            L.check(L.lib().octseg_plan_params_changed(plan.handle))
            plan.seen_version = ver
        if not train:
            plan.seen_buffers = bver
        if train:
            plan.stem_frame = x        # octseg.h: `image` must outlive the backward (the stem weight gradient reads it again)
        if train and self.encoder_name.startswith('efficientnet-'):
            plan.dc_factors = self._drop_connect_factors(plan, B, x.device)     # kept alive with the plan: the backward reads it too
            L.check(L.lib().octseg_plan_set_drop_connect(plan.handle, L.ptr(plan.dc_factors)))
        if self._has_dropout() and train:
            keep = self._draw_keep(B, H, W, x.device)
            plan.drop_keep = keep      # the backward of this step reads it too: keep it alive with the plan
            L.check(L.lib().octseg_plan_set_dropout(plan.handle, L.ptr(keep)))
        m = (C.c_float * 3)(*([float(v) for v in mean] if normalize else [0, 0, 0]))
        s = (C.c_float * 3)(*([float(v) for v in std] if normalize else [1, 1, 1]))
        want_graph = bool(self.use_graph and not train)
        if getattr(plan, 'graph_on', False) != want_graph and (want_graph or getattr(plan, 'graph_io', None) is not None):
            # the plan captures every eval forward while its graph mode is on -- also one issued eagerly on another stream
            # (use_graph switched off again): keep the C-side switch in step with the Python one
            L.check(L.lib().octseg_plan_set_graph(plan.handle, 1 if want_graph else 0))
            plan.graph_on = want_graph
        if want_graph:
            # serving path: the plan replays a captured hipGraph as long as every pointer stays the same, so the
            # frame goes through a persistent input buffer and the logits come back as a copy of a persistent one
            if getattr(plan, 'graph_io', None) is None:
                plan.graph_io = (torch.empty_like(x), torch.empty((B, self.classes, H, W), dtype=torch.float32, device=x.device),
                                 torch.cuda.Stream(device=x.device))   # the legacy default stream cannot be captured
            gin, gout, gstream = plan.graph_io
            cur = torch.cuda.current_stream(x.device)
            if getattr(plan, 'graph_pending', False):
                # an earlier forward_async of this (net, shape) was never joined: its replay may still read gin / write gout
                cur.wait_stream(gstream)
                plan.graph_pending = False
                warnings.warn('forward_async called again before forward_join: the earlier handle now aliases the new logits')
            gin.copy_(x)
            gstream.wait_stream(cur)
            with torch.cuda.stream(gstream):
                L.check(L.lib().octseg_net_forward(plan.handle, L.ptr(self.arena.data), L.ptr(self.bn_buffers),
                                                   L.ptr(plan.ws(x.device)), L.ptr(gin), L.ptr(gout), int(bool(normalize)), m, s, 0,
                                                   L.stream_ptr()))
            if defer_join:               # forward_async: the caller joins later, other nets' replays run beside this one
                plan.graph_pending = True
                return (gout, gstream, plan), plan
            cur.wait_stream(gstream)
            return gout.clone(), plan
        if defer_join:
            raise RuntimeError('forward_async needs eval mode and use_graph=True (the replay runs on the plan\'s own stream)')
        logits = torch.empty((B, self.classes, H, W), dtype=torch.float32, device=x.device)
        L.check(L.lib().octseg_net_forward(plan.handle, L.ptr(self.arena.data), L.ptr(self.bn_buffers), L.ptr(plan.ws(x.device)),
                                           L.ptr(x), L.ptr(logits), int(bool(normalize)), m, s, int(bool(train)),
                                           L.stream_ptr()))
        if train:
            self.num_batches_tracked += 1
            self._buffer_epoch += 1   # running_mean / running_var were rewritten behind torch's back
        return logits, plan

    def forward(self, x, normalize=False, mean=None, std=None):
        logits, _ = self._run_forward(x, normalize, mean, std, self.training)
        return logits

    def forward_async(self, x, normalize=False, mean=None, std=None):
        """Serving (eval, use_graph=True): enqueue this net's replayed forward on the plan's own stream and return at once.  Several nets
        started this way run side by side -- at one frame per step their grids fill a fraction of the chip each (the three-net ensemble
        of src/predict.py:61-101).  `forward_join(handle)` makes the current stream wait and returns the logits."""
        if self.training:
            raise RuntimeError('forward_async is a serving call: switch the net to eval()')
        handle, _ = self._run_forward(x, normalize, mean, std, False, defer_join=True)
        return handle

    @staticmethod
    def forward_join(handle):
        gout, gstream, plan = handle
        torch.cuda.current_stream(gout.device).wait_stream(gstream)
        plan.graph_pending = False
        return gout.clone()

    def dice(self, plan, logits, target):
        target = target.contiguous()
        if tuple(target.shape) != tuple(logits.shape) or target.dtype != torch.float32:
            raise ValueError(f'mask must be float32 {tuple(logits.shape)}, got {target.dtype} {tuple(target.shape)}')
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        stats = torch.empty((logits.shape[0], self.classes, 4), dtype=torch.int64, device=logits.device)
        L.check(L.lib().octseg_dice_forward(plan.handle, L.ptr(plan.ws(logits.device)), L.ptr(logits), L.ptr(target),
                                            L.ptr(loss), L.ptr(stats), L.stream_ptr()))
        return loss, stats

    def _forward_loss(self, image, target, normalize, mean, std):
        logits, plan = self._run_forward(image, normalize, mean, std, self.training)
        loss, stats = self.dice(plan, logits, target)
        return logits, loss, stats, plan

    def _backward(self, plan, logits, target, grad_scale=1.0, generation=None):
        if generation is not None and generation != plan.generation:
            raise RuntimeError(
                f'backward of a stale step: another forward of shape {plan.shape} ran on this network since the training_step '
                f'whose loss is being differentiated (its saved activations are gone). Call loss.backward() before the next '
                f'forward / predict of the same shape.')
        L.check(L.lib().octseg_net_backward(plan.handle, L.ptr(self.arena.data), L.ptr(self._grad_arena),
                                            L.ptr(plan.ws(logits.device)), L.ptr(logits), L.ptr(target.contiguous()),
                                            float(grad_scale), L.stream_ptr()))
        return self._grad_arena

    def dice_step(self, image, target, normalize=False, mean=None, std=None):
        """Forward + Dice loss.  Returns (loss, logits, stats[B,C,4]); ``loss.backward()`` runs the
        HIP backward and accumulates into ``self.arena.grad``."""
        if torch.is_grad_enabled() and self.training:
            return _DiceStep.apply(self.arena, self, image, target, normalize, mean, std)
        logits, loss, stats, _ = self._forward_loss(image, target, normalize, mean, std)
        return loss, logits, stats

    def train_step_raw(self, image, target, normalize=False, mean=None, std=None, grad_scale=1.0, exchange=None):
        """Forward + Dice + backward without autograd: gradients land in ``grad_arena`` (and are
        exposed as ``arena.grad`` without a copy).  The bench / DP loop uses this.  ``exchange``: a
        ``parallel.GradientExchange`` -- the backward then hands the gradient arena out slice by slice and the
        all-reduce of each slice overlaps the rest of the backward; on return the gradients are the cross-rank sums."""
        if self.use_train_graph and exchange is None and self.training:
            loss, logits, stats = self._graph_train_step(image, target, normalize, mean, std, grad_scale)
            self.arena.grad = self._grad_arena
            return loss, logits, stats
        logits, loss, stats, plan = self._forward_loss(image, target, normalize, mean, std)
        if exchange is not None:
            exchange.backward(plan, logits, target, grad_scale, generation=plan.generation)
        else:
            self._backward(plan, logits, target, grad_scale, generation=plan.generation)
        self.arena.grad = self._grad_arena
        return loss, logits, stats


def create_model(arch, encoder_name='resnet34', encoder_weights=None, in_channels=3, classes=1, **kwargs):
    """Drop-in for ``smp.create_model`` on the hot-path architectures."""
    return SegNet(arch, encoder_name, encoder_weights=encoder_weights, in_channels=in_channels, classes=classes, **kwargs)


def debug_tensor(net, plan, conv_name, grad=False):
    """Test hook: copy of a conv layer's raw output (or of its gradient buffer) as NCHW float32."""
    act, gr = C.c_size_t(), C.c_size_t()
    dims = (C.c_int * 4)()
    L.check(L.lib().octseg_plan_find_tensor(plan.handle, conv_name.encode(), C.byref(act), C.byref(gr), dims))
    N, H, W, Cc = list(dims)
    esz = 4 if net.dtype_code == L.F32 else 2
    off = gr.value if grad else act.value
    raw = plan.workspace[off:off + N * H * W * Cc * esz]
    t = raw.view(torch.float32 if esz == 4 else (torch.float16 if net.dtype_code == L.F16 else torch.bfloat16)).view(N, H, W, Cc)
    return t.float().permute(0, 3, 1, 2).contiguous()
