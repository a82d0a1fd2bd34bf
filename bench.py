#!/usr/bin/env python
"""bench.py -- OCT frames/s of the training hot path on MI355X.

One "step" = one optimisation step over a batch of synthetic OCT-shaped 704x704 frames:
forward (normalise + encoder-decoder) + Dice loss + backward + gradient all-reduce (N > 1)
+ fused Adam step, all inside the timed region, inputs resident in HBM.

Workload (BASELINE.json configs[1]): U-Net++ / resnet101, 1 class (Lumen), 704x704, bf16
storage + bf16 MFMA inputs with f32 accumulate, batch 16 per GPU.  N > 1: one process per GPU
(torchrun), data parallel with per-rank BN statistics / per-rank Dice and the RCCL all-reduce of
the flat gradient arena issued in slices beside the backward (Lightning-DDP semantics, reference
src/models/smp/train.py:122-133); per-GPU batch stays 16, so scaling is "weak" (`--scaling strong`
keeps the global batch at 16 instead: 16/N frames per GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline      -- the MFMA conv kernels (conv_mfma_kernel + gemm1x1_kernel + wgrad_mfma_kernel), timed live with HIP
                   events on the launch stream: algorithmic conv FLOPs (6 * MACs * frames, SURVEY.md
                   section 8d) / summed kernel time, vs 2.5 PFLOP/s dense bf16.  `achieved` comes from an
                   untimed extra pass with every launch on one stream (kernels alone); the brackets of the
                   timed, stream-overlapped steps are reported beside it under `overlapped`
  cpu_baseline  -- the torch-CPU oracle (a port, not the reference's own files) on the same workload at 704x704,
                   2 frames per step, 1 warm-up + 3 timed steps, median; fp32 (`value`) and bf16 autocast
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1 or '--force-exchange' in sys.argv:   # before HIP initialises: see oct_segmentation_amd/_lib.py
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')                                       # (side stream vs RCCL's streams; costs graph replay)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

WORKLOADS = {
    # name: (arch, encoder, classes, size)
    'unetpp_r101_704': ('unetplusplus', 'resnet101', 1, 704),
    'linknet_r50_704': ('linknet', 'resnet50', 2, 704),
    'unet_r50_704': ('unet', 'resnet50', 1, 704),
    'unet_r18_256': ('unet', 'resnet18', 1, 256),
    'fpn_r50_704': ('fpn', 'resnet50', 1, 704),       # sweep architectures outside BASELINE's three
    'deeplabv3plus_r50_704': ('deeplabv3plus', 'resnet50', 1, 704),
    'pspnet_r50_704': ('pspnet', 'resnet50', 1, 704),
    'deeplabv3_r50_704': ('deeplabv3', 'resnet50', 1, 704),
    'unet_regnetx064_704': ('unet', 'timm-regnetx_064', 1, 704),     # timm RegNet encoders of the sweep (configs/tune.yaml:19-24)
    'fpn_regnetx002_704': ('fpn', 'timm-regnetx_002', 1, 704),
    'unet_regnety120_704': ('unet', 'timm-regnety_120', 1, 704),
    'unet_effb0_704': ('unet', 'efficientnet-b0', 1, 704),           # efficientnet_pytorch encoders of the sweep (configs/tune.yaml:25-28)
    'fpn_effb5_704': ('fpn', 'efficientnet-b5', 1, 704),
    'manet_r50_704': ('manet', 'resnet50', 1, 704),                   # smp MAnet (configs/tune.yaml:17)
    'pan_r50_704': ('pan', 'resnet50', 1, 704),                       # smp PAN (configs/tune.yaml:18)
}


class PowerSampler:
    """Socket power and shader clock of one GPU, read from its hwmon files every 20 ms by a thread while the timed steps run (the conv
    loops sit on the package power limit on real operands -- profiles/r2_power_probe.txt -- so the clock they are given is part of
    reading `roofline.frac`).  Best effort: every field is None where the box does not expose the files."""

    def __init__(self, dev):
        import glob
        self.power = self.freq = self.cap = None
        self.samples = []
        dirs = []
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            dirs = glob.glob(f'/sys/bus/pci/devices/{bdf}/hwmon/hwmon*')
        except Exception:
            pass
        if not dirs:
            dirs = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))[:1]
        for d in dirs:
            for name in ('power1_average', 'power1_input'):
                if self.power is None and os.path.exists(os.path.join(d, name)):
                    self.power = os.path.join(d, name)
            if os.path.exists(os.path.join(d, 'freq1_input')):
                self.freq = os.path.join(d, 'freq1_input')
            if os.path.exists(os.path.join(d, 'power1_cap')):
                self.cap = os.path.join(d, 'power1_cap')
        self._stop = False
        self._thread = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            self.samples.append((self._read(self.power) if self.power else None, self._read(self.freq) if self.freq else None))
            time.sleep(0.02)

    def start(self):
        import threading
        if self.power or self.freq:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
        pw = [p for p, _ in self.samples if p is not None]
        fq = [f for _, f in self.samples if f is not None]
        cap = self._read(self.cap) if self.cap else None
        return {'socket_w': round(sum(pw) / len(pw) / 1e6, 1) if pw else None,
                'sclk_mhz': round(sum(fq) / len(fq) / 1e6, 1) if fq else None,
                'cap_w': round(cap / 1e6, 1) if cap else None, 'samples': len(self.samples),
                'note': 'hwmon power1_average / freq1_input of this GPU, sampled every 20 ms over the timed steps (rank 0)'}


def host_cores():
    """CPU threads this job may really use: the affinity mask, cut down to the cgroup CPU quota when there is one.  A one-GPU job
    on the GPU boxes sees every core of the host in its affinity mask but is scheduled on a 16-core share; running oneDNN with
    one thread per visible core then oversubscribes the share by an order of magnitude (a 704x704 step did not finish in 7
    minutes), hence the cap of 16 when the mask is large and no quota can be read."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = 'sched_getaffinity'
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:              # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()
            if q != 'max':
                quota = max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:   # cgroup v1
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, q // p)
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, how = quota, 'cgroup cpu quota'
    elif n > 32:
        n, how = 16, f'capped: affinity mask shows {n} cores, the one-GPU share of a box is 16'
    return n, how


def cpu_baseline(arch, enc, classes, size, batch=2, timed=3):
    """The CPU path beside the GPU number: the oracle's fwd + Dice + bwd + Adam
    step on the host cores at the benchmark's full frame size, batch 2 (frames/s is per frame; 16 frames of U-Net++/r101
    do not fit a sensible CPU budget), 1 warm-up + `timed` timed steps, median -- in fp32 (what the reference computes in)
    and under torch's bf16 autocast.  Bounded sample: ~8 steps of 2 frames."""
    from oracle import create_model, DiceLoss
    from synth import make_batch
    cores, how = host_cores()
    print(f'[bench] cpu baseline on {cores} threads ({how})', file=sys.stderr, flush=True)
    torch.set_num_threads(cores)
    loss_fn = DiceLoss()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img, mask = make_batch(batch, classes, size, seed=2)

    def run(autocast, budget_s):
        """1 warm-up + `timed` steps; stops early (and says so) when the leg exceeds its time budget -- bf16 convolutions have no fast
        path on every host CPU, and the default bench run has to finish within minutes."""
        torch.manual_seed(0)
        m = create_model(arch, enc, classes=classes).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-5)
        times, t_leg = [], time.time()
        for i in range(1 + timed):
            t0 = time.time()
            opt.zero_grad()
            if autocast:
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    z = m((img - mean) / std)
                loss = loss_fn(z.float(), mask)
            else:
                loss = loss_fn(m((img - mean) / std), mask)
            loss.backward()
            opt.step()
            times.append(time.time() - t0)
            print(f'[bench] cpu {"bf16-autocast" if autocast else "fp32"} step {i}: {times[-1]:.1f} s', file=sys.stderr, flush=True)
            if time.time() - t_leg + times[-1] > budget_s and i < timed:
                break
        kept = sorted(times[1:]) if len(times) > 1 else times     # the warm-up step only counts when it is all there is
        return kept[len(kept) // 2], times

    med32, t32 = run(False, 120.0)
    med16, t16 = run(True, 60.0)
    return {'value': round(batch / med32, 5), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'bf16_autocast_value': round(batch / med16, 5),
            'sample': f'{batch} frames {size}x{size} per step ({arch}/{enc}, fwd+Dice+bwd+Adam), torch {torch.__version__} CPU oracle on {cores} '
                      f'threads; step times incl. the warm-up step 0 (median of the rest, or the warm-up alone if the leg ran out of its '
                      f'time budget): fp32 {[round(t, 2) for t in t32]} s, bf16 autocast {[round(t, 2) for t in t16]} s'}


_RESULT_OUT = sys.stdout

ENSEMBLE = (('LM', 'unetplusplus', 'resnet101', ['Lumen']), ('FC_LC', 'linknet', 'resnet50', ['Lipid core', 'Fibrous cap']),
            ('VV', 'unet', 'resnet50', ['Vasa vasorum']))


def bench_ensemble(args):
    """BASELINE config #5: the predict.py path -- LM (U-Net++/resnet101) + FC_LC (LinkNet/resnet50, 2 classes) + VV (U-Net/resnet50)
    eval forwards in fp16 on `--batch` 704x704 frames (no normalisation, reference model.py:192), each net once per batch (the
    reference runs FC_LC twice), sigmoid > 0.5 + cv2-nearest resize to 1000x1000 + 4-channel mask assembly on the GPU
    (octseg_mask_assemble, predict.py:92-100).  One step = one batch through all three nets and the epilogue; frames resident in
    HBM.  Replayed hipGraphs unless --no-graph."""
    import ctypes as C
    from oct_segmentation_amd import _lib as L
    if os.environ.get('OCTSEG_LIB'):   # A/B of two builds of the library on one box (tools/ab_perf.sh); the default is the in-tree build
        L.LIB_PATH = os.path.abspath(os.environ['OCTSEG_LIB'])
    from oct_segmentation_amd.engine import SegNet
    from oct_segmentation_amd.model import CLASS_IDS
    from oct_segmentation_amd.predict import MODELS_META, cv2_nearest_index
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    B, S, OUT = args.batch, 704, 1000
    cdt = {'fp16': torch.float16, 'bf16': torch.bfloat16, 'fp32': torch.float32}[args.dtype]
    nets = []
    for i, (_, arch, enc, classes) in enumerate(ENSEMBLE):
        net = SegNet(arch, enc, classes=len(classes), device=dev, compute_dtype=cdt, seed=40 + i).eval()
        net.use_graph = not args.no_graph
        nets.append((net, classes))
    from synth import make_batch
    x = make_batch(B, 1, S, seed=7)[0].to(dev)
    stack = torch.zeros((B, OUT, OUT, 4), dtype=torch.float32, device=dev)
    rows = torch.from_numpy(cv2_nearest_index(S, OUT)).to(dev)
    lib = L.lib()

    side_by_side = not args.no_graph and not args.serial_nets

    def step():
        # replayed graphs of the three nets are started together, each on its plan's own stream, and joined in order (SegNet.forward_async)
        handles = [net.forward_async(x, normalize=False) if (side_by_side and net.use_graph) else None for net, _ in nets]
        for (net, classes), h in zip(nets, handles):
            if os.environ.get('OCTSEG_BENCH_TRACE'):
                print(f'[bench] {net.arch}/{net.encoder_name}', file=sys.stderr, flush=True)
            z = net.forward_join(h) if h is not None else net(x, normalize=False)
            for cl in classes:
                ch = MODELS_META[cl]['index'] if z.shape[1] > 1 else 0
                L.check(lib.octseg_mask_assemble(L.ptr(z), B, z.shape[1], S, S, int(ch), L.ptr(stack), OUT, OUT, 4, CLASS_IDS[cl] - 1,
                                                 L.ptr(rows), L.ptr(rows), L.stream_ptr()))

    print(f'[bench] ensemble of 3 nets, {B} frame(s) per step, {args.dtype}, {"hipGraph replay" if not args.no_graph else "eager"}: warm-up',
          file=sys.stderr, flush=True)
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # roofline pass: eager, one stream, every conv launch bracketed by HIP events
    for net, _ in nets:
        net.use_graph = False
    prof = (C.c_double * 12)()
    L.check(lib.octseg_debug_set_serial(1))
    step(); torch.cuda.synchronize()
    L.check(lib.octseg_profile_start())
    n_alone = 3
    for _ in range(n_alone):
        step()
    torch.cuda.synchronize()
    L.check(lib.octseg_profile_stop(prof))
    L.check(lib.octseg_debug_set_serial(0))
    macs = sum(net.fwd_macs(B, S, S) for net, _ in nets) / B
    ach = prof[1] / (prof[0] * 1e-3) / 1e12 if prof[0] > 0 else 0.0
    peak = 157.3 if args.dtype == 'fp32' else 2500.0
    out = {'metric': f'OCT frames/sec (704x704, {args.dtype}) 3-net ensemble inference', 'value': round(B * args.steps / dt, 3), 'unit': 'frames/s',
           'n_gpus': 1, 'steps': args.steps, 'warmup': max(args.warmup, 3), 'ms_per_step': round(dt / args.steps * 1e3, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
           'data': 'synthetic OCT-shaped frames (seeded), random-init weights',
           'config': {'multi_gpu': 'replicas only: frames are independent, this workload runs on one GPU', 'fc_lc_forwards_per_batch': 1,   # the reference runs the FC_LC net once per class = twice (predict.py:70-76)
                      'workload': f'ensemble LM unetplusplus/resnet101 + FC_LC linknet/resnet50 (2 classes) + VV unet/resnet50, 704x704 -> 4-class '
                                  f'1000x1000 masks, batch {B}, eval forward + GPU mask assembly, {"hipGraph replay" if not args.no_graph else "eager launches"}',
                      'global_batch': B, 'parallelism': 'dp1', 'gmac_fwd_per_frame': round(macs / 1e9, 2)},
           'roofline': {'bound': 'mfma', 'kernel': 'conv_mfma_kernel + gemm1x1_kernel (forward, BatchNorm folded)', 'achieved': round(ach, 2),
                        'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': None,
                        'launches_per_step': round(prof[2] / n_alone, 1), 'kernel_ms_per_step': round(prof[0] / n_alone, 3),
                        'avg_launch_ms': round(prof[0] / max(prof[2], 1.0), 4), 'algorithmic_gflop_per_frame': round(2 * macs / 1e9, 1),
                        'note': 'HIP-event brackets of every conv launch in an untimed eager one-stream pass of the same step'}}
    print(f'[bench] ensemble: {out["value"]} frames/s, {out["ms_per_step"]} ms/step; conv kernels alone {ach:.1f} TFLOP/s', file=sys.stderr, flush=True)
    print(json.dumps(out), file=_RESULT_OUT, flush=True)


def main():
    # stdout carries exactly ONE line, the JSON: libraries that print banners to file descriptor 1 (RCCL writes its version block there when
    # a communicator is created) are sent to stderr for the life of the process; the result goes to the saved descriptor
    global _RESULT_OUT
    sys.stdout.flush()
    _RESULT_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='unetpp_r101_704', choices=sorted(WORKLOADS) + ['ensemble_704_fp16'])
    ap.add_argument('--no-graph', action='store_true', help='ensemble workload: eager launches instead of replayed hipGraphs')
    ap.add_argument('--serial-nets', action='store_true', help='ensemble workload: one net after the other instead of the three replays side by side')
    ap.add_argument('--batch', type=int, default=16, help='frames per GPU (weak scaling) / frames in the global batch (strong scaling)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: --batch frames per GPU whatever N; strong: the global batch stays --batch, every GPU gets batch/N frames')
    ap.add_argument('--allreduce-slices', type=int, default=3, help='gradient-arena slices all-reduced beside the backward (N > 1)')
    ap.add_argument('--allreduce-dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='dtype of the gradients on the wire (N > 1): bf16 halves the bytes of the per-link-bound xGMI ring')
    ap.add_argument('--loss', default='dice', choices=['dice', 'bce', 'dice+bce'], help="criterion (the reference's is Dice, model.py:55)")
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp32', 'fp16'], help='default: bf16 (training workloads), fp16 (ensemble)')
    ap.add_argument('--optimizer', default='Adam')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)')
    ap.add_argument('--profile-timed', action='store_true', help='also bracket every launch of the timed steps with HIP events (roofline.overlapped)')
    ap.add_argument('--train-graph', action='store_true',
                    help='forward + Dice + backward of a step as one replayed hipGraph (octseg_net_train_step); N = 1 or without the exchange')
    ap.add_argument('--force-exchange', action='store_true',
                    help='N = 1: run the data-parallel step anyway -- a one-rank RCCL group, buffer broadcast and the sliced all-reduce '
                         'beside the backward (exercises the nccl branch on a single GPU; the collectives are identities)')
    args = ap.parse_args()
    if args.workload == 'ensemble_704_fp16':
        args.dtype = args.dtype or 'fp16'
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: no GPU visible (there is no CPU fallback)')
        if args.batch == 16 and '--batch' not in ' '.join(sys.argv):
            args.batch = 1     # the reference's predict loop is frame by frame (predict.py:85-91)
        return bench_ensemble(args)
    args.dtype = args.dtype or 'bf16'
    if args.dtype == 'fp16':
        raise SystemExit('fp16 is the serving dtype (--workload ensemble_704_fp16); training runs in bf16 or fp32')

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no GPU visible (there is no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU; wraps only in the single-GPU gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dp = world > 1 or args.force_exchange      # the data-parallel step (process group, buffer broadcast, sliced all-reduce)
    if dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', '29531')
        kw = dict(rank=rank, world_size=world) if world == 1 else {}
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, **kw)
        else:
            dist.init_process_group(args.backend, **kw)

    from oct_segmentation_amd import _lib as L
    if os.environ.get('OCTSEG_LIB'):   # A/B of two builds of the library on one box (tools/ab_perf.sh); the default is the in-tree build
        L.LIB_PATH = os.path.abspath(os.environ['OCTSEG_LIB'])
    from oct_segmentation_amd.model import OCTSegmentationModel
    from oct_segmentation_amd.parallel import (GradientExchange, GradientExchangeError, broadcast_buffers, broadcast_parameters,
                                               exit_on_exchange_failure, shard_range)
    from synth import make_batch
    import ctypes as C

    arch, enc, classes, S = WORKLOADS[args.workload]
    B = args.batch
    if args.scaling == 'strong':   # fixed global batch, sharded like the data-parallel loader does
        lo, hi = shard_range(args.batch, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit(f'--scaling strong: global batch {args.batch} leaves rank {rank} of {world} without a frame')
    global_batch = args.batch if args.scaling == 'strong' else world * B
    cdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    names = ['Lumen', 'Fibrous cap', 'Lipid core', 'Vasa vasorum'][:classes]
    model = OCTSegmentationModel(arch, enc, 'bench', 3, names, lr=1e-5, weight_decay=0.0, optimizer_name=args.optimizer,
                                 input_size=S, device=dev, compute_dtype=cdt, seed=1234, loss=args.loss)
    model.train()
    net = model.model
    net.use_train_graph = bool(args.train_graph)
    exchange = None
    # diagnostics of the data-parallel step's own cost on one GPU (--force-exchange): leave one of its two parts out
    no_bcast, no_exch = bool(os.environ.get('OCTSEG_BENCH_NO_BCAST')), bool(os.environ.get('OCTSEG_BENCH_NO_EXCHANGE'))
    if dp:
        broadcast_parameters(net)
        if not no_exch:
            exchange = GradientExchange(net, nslices=args.allreduce_slices, wire_dtype=args.allreduce_dtype)
    opt = model.configure_optimizers()
    img, mask = make_batch(B, classes, S, seed=1234 + rank)
    img, mask = img.to(dev), mask.to(dev)

    def step():
        if dp and not no_bcast:
            broadcast_buffers(net)  # torch-DDP broadcast_buffers=True
        # grad_scale 1/world + SUM all-reduce == DDP's gradient mean; the all-reduce runs slice by slice beside the backward
        try:
            loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=model._mean, std=model._std,
                                                     grad_scale=1.0 / world, exchange=exchange)
        except GradientExchangeError as e:     # the process group is aborted: leave non-zero, torchrun takes the job down
            exit_on_exchange_failure(e)
        opt.step()
        return loss

    def barrier():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f'[bench] {msg}', file=sys.stderr, flush=True)

    note(f'{arch}/{enc} {S}x{S} batch {B}/GPU {args.dtype}: warm-up x{args.warmup}')
    for _ in range(args.warmup):
        step()
    barrier()
    note(f'timing {args.steps} steps')
    # HIP-event brackets around every launch of the TIMED steps cost the host two event records per launch (~800 per step): opt-in
    if args.profile_timed:
        L.check(L.lib().octseg_profile_start())
    sampler = PowerSampler(dev) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    enq = 0.0
    for _ in range(args.steps):
        te = time.perf_counter()
        loss = step()
        enq += time.perf_counter() - te      # host time to ENQUEUE a step (no synchronisation inside): how far the host runs ahead
    barrier()
    dt = time.perf_counter() - t0
    power = sampler.stop() if sampler is not None else None
    prof = (C.c_double * 12)()
    if args.profile_timed:
        L.check(L.lib().octseg_profile_stop(prof))
    loss_val = float(loss.item())
    # Roofline pass (untimed, after the measured steps): the same step with every launch on one stream, so that the
    # HIP-event bracket of a launch is the duration of that kernel alone.  In the timed steps the weight gradients and
    # part of the decoder run on a side stream; brackets taken there also contain the time a kernel shares the chip.
    prof_alone = (C.c_double * 12)()
    n_alone = 2
    net.use_train_graph = False          # the roofline pass brackets every launch: eager, one stream
    L.check(L.lib().octseg_debug_set_serial(1))
    step()
    barrier()
    L.check(L.lib().octseg_profile_start())
    for _ in range(n_alone):
        step()
    barrier()
    L.check(L.lib().octseg_profile_stop(prof_alone))
    L.check(L.lib().octseg_debug_set_serial(0))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames = global_batch * args.steps
        macs = net.fwd_macs(B, S, S) / B  # per frame
        oms = [prof[0], prof[3], prof[6]]          # overlapped (timed) steps
        ofl = [prof[1], prof[4], prof[7]]
        ms = [prof_alone[0], prof_alone[3], prof_alone[6]]   # kernels alone (roofline pass)
        fl = [prof_alone[1], prof_alone[4], prof_alone[7]]
        nl = [prof_alone[2], prof_alone[5], prof_alone[8]]
        tot_ms, tot_fl = sum(ms), sum(fl)
        peak = 2500.0 if args.dtype == 'bf16' else 157.3
        ach = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        ex = [e / B for e in net.exec_macs(B, S, S)]          # per frame: forward, data gradient, weight gradient
        ex_fl = tot_fl - n_alone * B * 2.0 * sum(macs - e for e in ex)          # FLOPs the bracketed launches executed
        ach_ex = ex_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
        executed = {'gflop_per_frame': round(2 * sum(ex) / 1e9, 1), 'share_of_algorithmic': round(sum(ex) / (3 * macs), 4) if macs > 0 else 1.0,
                    'achieved': round(ach_ex, 2), 'frac': round(ach_ex / peak, 4),
                    'by_class_gflop_per_frame': {k: round(2 * e / 1e9, 1) for k, e in zip(('fwd', 'dgrad', 'wgrad'), ex)}}
        out = {
            'metric': f'OCT frames/sec ({S}x{S}, {args.dtype}) fwd+bwd',
            'value': round(frames / dt, 3),
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic OCT-shaped frames (seeded), random-init weights',
            'config': {'workload': f'{arch}/{enc} {classes}-class {S}x{S}, batch {B}/GPU, fwd+{ {"dice": "Dice", "bce": "BCE", "dice+bce": "Dice+BCE"}[args.loss] }+bwd+allreduce{"(bf16)" if args.allreduce_dtype == "bf16" else ""}+{args.optimizer}',
                       'global_batch': global_batch, 'parallelism': f'dp{world}', 'gmac_fwd_per_frame': round(macs / 1e9, 2)},
            'loss': round(loss_val, 6),
            'host_enqueue_ms_per_step': round(enq / args.steps * 1e3, 3),
            'power': power,
            'roofline': {
                'bound': 'mfma', 'kernel': 'conv_mfma_kernel + conv3x3p_kernel + gemm1x1_kernel + thin_conv_kernel + wgrad_mfma_kernel + wgrad1x1_kernel + wgrad_convt16_kernel + thin_wgrad_kernel (implicit-GEMM conv fwd / dgrad / wgrad)',
                'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': None,
                'launches_per_step': round(sum(nl) / n_alone, 1),
                'avg_launch_ms': round(tot_ms / max(1.0, sum(nl)), 4),
                'kernel_ms_per_step': round(tot_ms / n_alone, 3),
                'by_class': {k: {'ms_per_step': round(m / n_alone, 3), 'tflops': round(f / (m * 1e-3) / 1e12, 2) if m > 0 else 0.0}
                             for k, m, f in zip(('fwd', 'dgrad', 'wgrad'), ms, fl)},
                'algorithmic_gflop_per_frame': round(6 * macs / 1e9, 1),
                # `achieved` prices the reference graph's 6 x MACs.  Where the plan runs the decoder's nearest-x2 + 3x3 layers as a 4x4 stride-2
                # kernel over the low-resolution map (same function of the same weights, 16 instead of 36 products per source pixel:
                # octseg_plan_exec_macs) the matrix cores execute fewer FLOPs than that -- both are printed
                'executed': executed,
                'note': 'HIP-event brackets on the launch stream over an untimed pass of the same step with every launch on one '
                        'stream (octseg_debug_set_serial): the duration of each kernel alone; rocprofv3 summary of that mode: '
                        'profiles/r3_serial_kernel_stats.csv',
                'overlapped': None if not args.profile_timed else {
                    'note': 'the same brackets during the TIMED steps, where weight gradients and part of the decoder run on a '
                            'side stream: durations include the time a kernel shares the chip (profiles/r3_bench_kernel_stats.csv)',
                    'kernel_ms_per_step': round(sum(oms) / args.steps, 3),
                    'achieved': round(sum(ofl) / (sum(oms) * 1e-3) / 1e12, 2) if sum(oms) > 0 else 0.0,
                    'by_class': {k: {'ms_per_step': round(m / args.steps, 3), 'tflops': round(f / (m * 1e-3) / 1e12, 2) if m > 0 else 0.0}
                                 for k, m, f in zip(('fwd', 'dgrad', 'wgrad'), oms, ofl)},
                },
            },
        }
        if power and power.get('sclk_mhz'):   # the clock the card granted during the timed steps (sampled, not assumed)
            mhz = float(power['sclk_mhz'])
            out['roofline']['granted_clock'] = {
                'sclk_mhz': mhz, 'quoted_mhz': 2400, 'frac_at_granted_clock': round(ach / (peak * mhz / 2400.0), 4),
                'note': 'hwmon freq1_input averaged over the timed steps of THIS run (a step alternates power-capped conv loops with HBM '
                        'sweeps, so the conv loops themselves run below this average); `peak` is the 2.4 GHz figure'}
        if args.train_graph:
            out['config']['train_graph'] = 'forward + Dice + backward replayed as one hipGraph per step (octseg_net_train_step)'
        if args.force_exchange:
            out['config']['force_exchange'] = f'one-rank {args.backend} group: buffer broadcast + {args.allreduce_slices}-slice all-reduce issued beside the backward'
        # HBM traffic of the MFMA kernels from the TCC counters (tools/collect_traffic.py, separate rocprofv3 --pmc passes)
        tpath = next((q for q in (os.path.join(ROOT, 'profiles', f'r{k}_traffic.json') for k in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)), '')
        if os.path.exists(tpath) and args.workload == 'unetpp_r101_704' and B == 16 and args.dtype == 'bf16':
            try:
                out['roofline']['traffic'] = round(json.load(open(tpath))['mfma_family']['hbm_bytes_per_launch'])
                out['roofline']['traffic_unit'] = f'HBM bytes per launch (FETCH_SIZE*2 + WRITE_SIZE, profiles/{os.path.basename(tpath)})'
            except Exception:
                pass
        # the HBM-bound part of the step (SURVEY.md section 8d asks for both roofs): BatchNorm sweeps, kernels alone
        if prof_alone[9] > 0:
            gbs = prof_alone[10] / (prof_alone[9] * 1e-3) / 1e9
            out['roofline_hbm'] = {
                'bound': 'hbm', 'kernel': 'bn_bwd_apply + bn_bwd_reduce + bn_act (NHWC sweeps, 16-byte vectors)',
                'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                'kernel_ms_per_step': round(prof_alone[9] / n_alone, 3), 'launches_per_step': round(prof_alone[11] / n_alone, 1),
                'algorithmic_gbytes_per_step': round(prof_alone[10] / n_alone / 1e9, 2),
                'note': 'algorithmic bytes = every tensor these sweeps read or write, once; same untimed single-stream pass as roofline'}
        note(f'GPU: {out["value"]} frames/s, {out["ms_per_step"]} ms/step; MFMA kernels {ach:.1f} TFLOP/s')
        if not args.no_cpu_baseline and world == 1:
            note('CPU baseline (oracle, 2 frames per step, fp32 + bf16 autocast) ...')
            out['cpu_baseline'] = cpu_baseline(arch, enc, classes, S)
        print(json.dumps(out), file=_RESULT_OUT, flush=True)
    if dp:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
