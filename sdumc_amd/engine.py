"""Thin Python host over the network-level C ABI (include/sdumc_hip.h, engine.hip).

PyTorch-ROCm is used for device memory and streams only.  Everything numeric
happens in libsdumc_hip.so; there is no CPU or torch fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check, ptr

D, H, NQ, RNC_DIM = _lib.D, _lib.H, _lib.NQ, _lib.RNC_DIM
P_FRAME, P_MLP = 0.5, 0.3        # model :54,:77 / model :187 (constructor default; --dropout is never forwarded)
DEFAULT_WEIGHTS = (0.5, 0.5, 0.1, 0.7, 0.1, 0.8)      # main :234-239
PUBLISHED_WEIGHTS = (0.5, 0.5, 0.0, 0.0, 0.13, 0.5)   # shell/main_text_missing_icassp.sh:6


class ParamLayout:
    """name -> (offset, shape, live) of the flat parameter buffer for feature widths (da, dt, dv)."""

    _cache = {}

    def __init__(self, da, dt, dv):
        self.dims = (int(da), int(dt), int(dv))
        need = -lib.sdumc_param_table(*self.dims, None, 0)
        buf = C.create_string_buffer(need)
        n = lib.sdumc_param_table(*self.dims, buf, need)
        if n < 0:
            raise _lib.SdumcError("sdumc_param_table failed")
        self.entries = {}
        self.order = []
        for line in buf.value.decode().splitlines():
            name, off, rows, cols, live = line.split()
            shape = (int(rows), int(cols)) if int(cols) else (int(rows),)
            self.entries[name] = (int(off), shape, bool(int(live)))
            self.order.append(name)
        self.total = lib.sdumc_param_count(*self.dims)
        self.live = lib.sdumc_param_live_count(*self.dims)
        # [0, early): utterance-level layers (gradients final after backward phase 0), [early, live): frame-level
        self.early = lib.sdumc_param_early_count(*self.dims)

    @classmethod
    def get(cls, da, dt, dv):
        key = (int(da), int(dt), int(dv))
        if key not in cls._cache:
            cls._cache[key] = cls(*key)
        return cls._cache[key]

    def views(self, flat):
        """dict name -> view of `flat` (no copies)."""
        out = {}
        for name in self.order:
            off, shape, _ = self.entries[name]
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off:off + n].view(shape)
        return out

    def live_names(self):
        return [n for n in self.order if self.entries[n][2]]


def _require_cuda(*tensors, dtypes=(torch.float32,)):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.SdumcError("sdumc_amd runs on the GPU only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype not in dtypes or not t.is_contiguous():
            raise _lib.SdumcError("expected contiguous " + " / ".join(str(d).replace("torch.", "") for d in dtypes) + " tensors")


def bf16_mode(bf16, dims):
    """sdumc_net_dims.bf16 for the Python-level switch: False -> 0 (fp32); True -> 2 = bf16 STORAGE of features, projected
    frames, keys and frame-level gradients (BASELINE configs[2] / [4]) when the feature widths are whole 64-element k-tiles,
    else 1; "operands" -> 1 = fp32 storage, frame-level GEMM operands rounded to bf16 on their way to the matrix cores."""
    if not bf16:
        return 0
    if bf16 == "operands" or (not isinstance(bf16, bool) and bf16 == 1):
        return 1
    return 2 if all(int(d) % 64 == 0 for d in dims[:3]) else 1


def planes_wanted(planes, dims, bf16):
    """Whether a step / call keeps bf16-plane copies of its fp32 features (sdumc_net_io.*_p3; csrc/gemm_p3.hip: the frame and key
    projections then run on operands split once per tensor).  OPT-IN (planes=True): the caller states that the installed batch is
    RESIDENT -- run many times, or assembled from a DeviceFeatureStore(planes=True) whose planes were split once per dataset.  A loop
    that installs a fresh batch per step (set_batch per step) would re-split ~224 MB per step at C2 for one use,
    ~0.15-0.2 ms against the ~0.015 ms the planes save: there the default (False = the in-kernel split of csrc/gemm_wide.hip) is the
    faster path.  None: the environment's SDUMC_P3 (A/B runs), else False.  Needs fp32 storage and widths in whole 64-element k-tiles;
    costs 1.5x the features' bytes on top of them."""
    import os
    if planes is None:
        planes = os.environ.get("SDUMC_P3", "0") == "1"
    return bool(planes) and bf16_mode(bf16, dims) == 0 and all(int(d) % 64 == 0 for d in dims[:3])


def p3_split_into(src, dst):
    """src fp32 [..., d] contiguous -> dst uint8 [rows, 6 d]: the three bf16 planes of every value (exact: they sum to it)"""
    d = src.shape[-1]
    rows = src.numel() // d
    check(lib.sdumc_p3_split(ptr(src), d, ptr(dst), 6 * d, rows, d, _lib.current_stream()), "sdumc_p3_split")
    return dst


def make_dims(B, streams, Ta, Tv, Tt, dims, train, sample0=0, p_mlp=P_MLP, bf16=False):
    d = _lib.NetDims()
    d.B, d.streams, d.Ta, d.Tv = B, streams, Ta, Tv
    d.Tt[0] = Tt[0]
    d.Tt[1] = Tt[1] if len(Tt) > 1 else Tt[0]
    d.da, d.dt, d.dv = dims[0], dims[1], dims[2]
    d.train = 1 if train else 0
    d.sample0 = sample0
    d.p_frame, d.p_mlp = P_FRAME, p_mlp
    d.bf16 = bf16_mode(bf16, dims)
    return d


class ExecContext:
    """A caller-owned execution context of the C ABI (sdumc_ctx_create): its own internal side streams and event ring.
    Steps driven concurrently from several host threads (each on its own torch stream) take one context each; without one,
    every call on a device shares that device's default context and must come from one thread at a time."""

    def __init__(self):
        h = C.c_void_p()
        check(lib.sdumc_ctx_create(C.byref(h)), "sdumc_ctx_create")
        self.handle = h

    OPTIONS = {"concurrency": 0, "background_lane": 1, "chain_cluster": 2, "split": 3}

    def set_option(self, name, value):
        """Schedule options of THIS context only (sdumc_ctx_set_option): 'concurrency' 0 | 1, 'background_lane' 0 | 2 | 3,
        'chain_cluster' 0 | 1, 'split' 0..15 (which fp32 GEMM kernel families multiply on the bf16 matrix pipe: sdumc_hip.h,
        sdumc_set_split_); None (or a negative value) returns the option to the process-wide default."""
        check(lib.sdumc_ctx_set_option(self.handle, self.OPTIONS[name], -1 if value is None else int(value)), "sdumc_ctx_set_option")

    def close(self):
        if self.handle:
            lib.sdumc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RngState:
    """Device-resident {seed_lo, seed_hi, call}: lets captured graphs draw fresh masks per replay."""

    def __init__(self, seed, device, call=0):
        self.t = torch.tensor([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, call], dtype=torch.int64,
                              device="cpu").to(torch.int32).to(device)

    def set_call(self, call):
        self.t[2] = call

    @property
    def call(self):
        return int(self.t[2].item()) & 0xFFFFFFFF


def _lengths_arg(lengths, B, n, device):
    """(audio, text, video[, feat4]) valid-frame counts -> n contiguous cuda int32 tensors of B entries."""
    if lengths is None:
        return None
    lengths = list(lengths)
    if len(lengths) != n:
        raise _lib.SdumcError(f"lengths: expected {n} per-modality tensors (audio, text, video" + (", feat4)" if n == 4 else ")"))
    out = []
    for l in lengths:
        t = torch.as_tensor(l, dtype=torch.int32).reshape(-1).to(device).contiguous()
        if t.numel() != B:
            raise _lib.SdumcError("lengths: one entry per sample of the batch")
        out.append(t)
    return out


class NetCall:
    """One network invocation (1 or 2 streams): owns workspace + outputs, supports backward.
    `lengths` (extension, default None = the reference's behaviour): per-modality valid frame counts
    (audio, text[, feat4 when two streams], video -> given as (audio, text, video) or (audio, text, video, feat4));
    padded frames are then masked out of the six attention poolings."""

    def __init__(self, flat_params, audio, texts, video, train, rng, sample0=0, p_mlp=P_MLP, bf16=False, lengths=None,
                 ctx=None, planes=None, bits_next=False):
        texts = list(texts)
        _require_cuda(flat_params)
        _require_cuda(audio, video, *texts, dtypes=(torch.float32, torch.bfloat16))
        S = len(texts)
        B, Ta, da = audio.shape
        store = torch.bfloat16 if bf16_mode(bf16, (da, texts[0].shape[2], video.shape[2])) == 2 else torch.float32
        # bf16-storage mode reads bf16 features (a DeviceFeatureStore(bf16=True) hands them over as such; fp32 inputs are
        # converted once, here); the other modes read fp32
        audio, video = audio.to(store), video.to(store)
        texts = [t.to(store) for t in texts]
        Tv, dv = video.shape[1], video.shape[2]
        dt = texts[0].shape[2]
        for t in texts:
            if t.shape[0] != B or t.shape[2] != dt:
                raise _lib.SdumcError("text-slot inputs must share batch and width (SURVEY §8b: feat4 width == text width)")
        if video.shape[0] != B:
            raise _lib.SdumcError("batch mismatch")
        self.dims = make_dims(B, S, Ta, Tv, [t.shape[1] for t in texts], (da, dt, dv), train, sample0, p_mlp, bf16)
        self.layout = ParamLayout.get(da, dt, dv)
        if flat_params.numel() != self.layout.total:
            raise _lib.SdumcError("flat parameter buffer has the wrong size")
        dev = audio.device
        nbytes = lib.sdumc_net_workspace_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise _lib.SdumcError("invalid network dimensions")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        V = B * S
        self.V, self.B, self.S = V, B, S
        self.vals = torch.empty(V, 1, device=dev)
        self.fused = torch.empty(V, H, device=dev)
        self.rnc = torch.empty(V, RNC_DIM, device=dev)
        self.text_hidden = torch.empty(V, D, device=dev)
        self.cross_text = torch.empty(V, NQ, H, device=dev)
        self._keep = (flat_params, audio, texts, video, rng)
        io = _lib.NetIO()
        io.audio, io.video = ptr(audio), ptr(video)
        io.text[0] = ptr(texts[0])
        io.text[1] = ptr(texts[1]) if S == 2 else None
        io.params = ptr(flat_params)
        io.rng_state = ptr(rng.t) if rng is not None else None
        io.workspace, io.workspace_bytes = ptr(self.workspace), nbytes
        io.vals, io.fused, io.rnc = ptr(self.vals), ptr(self.fused), ptr(self.rnc)
        io.text_hidden, io.cross_text = ptr(self.text_hidden), ptr(self.cross_text)
        self._lengths = _lengths_arg(lengths, B, 4 if S == 2 else 3, dev)
        if self._lengths is not None:
            for i, t in enumerate(self._lengths):
                io.lengths[i] = ptr(t)
        self._ctx = ctx
        io.ctx = ctx.handle if ctx is not None else None
        self._planes = None
        if planes_wanted(planes, (da, dt, dv), bf16):      # bf16-plane copies of the features, split once for the life of this call object
            mk = lambda t: torch.empty(t.shape[0] * t.shape[1], 6 * t.shape[2], dtype=torch.uint8, device=dev)
            self._planes = (mk(audio), mk(video), [mk(t) for t in texts])
            io.audio_p3, io.video_p3 = ptr(self._planes[0]), ptr(self._planes[1])
            for i, t in enumerate(self._planes[2]):
                io.text_p3[i] = ptr(t)
        # fp32 train-mode calls that are repeated step after step on this object (the data-parallel backend): the NEXT call's frame-level
        # keep-bits are generated in this call's middle (sdumc_net_io.bits_next; the Philox counter must advance by 2 between calls,
        # as HipBackend.adam and sdumc_train_step do) -- bit-identical masks, tagged; any other sequence regenerates at the head
        self._bits_next = None
        if bits_next and train and rng is not None:
            nb = lib.sdumc_net_bits_next_bytes(C.byref(self.dims))
            if nb:
                self._bits_next = torch.zeros(nb, dtype=torch.uint8, device=dev)
                io.bits_next = ptr(self._bits_next)
        self.io = io
        self.refresh_planes()

    def refresh_planes(self):
        """Re-split the feature tensors this call object was built on (a caller that overwrites them in place -- the data-parallel
        backend's resident batch buffers -- calls this after every new batch)."""
        if self._planes is None:
            return
        _, audio, texts, video, _ = self._keep
        p3_split_into(audio, self._planes[0])
        p3_split_into(video, self._planes[1])
        for t, dst in zip(texts, self._planes[2]):
            p3_split_into(t, dst)

    def set_lengths(self, lengths):
        """Key-padding extension for the next forward: per-modality valid frame counts, or None = the reference's behaviour."""
        n = 4 if self.S == 2 else 3
        new = _lengths_arg(lengths, self.B, n, self.vals.device)
        if new is None:
            self._lengths = None
            for i in range(4):
                self.io.lengths[i] = None
            return
        if self._lengths is None:
            self._lengths = [torch.empty(self.B, dtype=torch.int32, device=self.vals.device) for _ in range(n)]
            for i, t in enumerate(self._lengths):
                self.io.lengths[i] = ptr(t)
        for dst, src in zip(self._lengths, new):
            dst.copy_(src, non_blocking=True)

    def forward(self):
        check(lib.sdumc_net_forward(C.byref(self.dims), C.byref(self.io), _lib.current_stream()), "sdumc_net_forward")
        self._phase_used = self.io.bits_phase      # (the backward of this call reads the set the forward read)
        return self.vals, self.fused, self.rnc, self.text_hidden, self.cross_text

    def next_call(self):
        """After the step that this forward / backward pair belongs to (the caller has advanced the Philox counter by 2): the next
        forward reads the keep-bits set this one filled."""
        if self._bits_next is not None:
            self.io.bits_phase ^= 1

    def backward(self, d_vals, d_fused, d_rnc, d_text_hidden, d_cross_text, grads=None):
        """Returns the flat gradient bucket [live] (allocated zeroed when not given)."""
        _require_cuda(d_vals, d_fused, d_rnc, d_text_hidden, d_cross_text)
        if grads is None:
            grads = torch.zeros(self.layout.live, device=self.vals.device)
        g = _lib.NetGrads()
        g.d_vals, g.d_fused, g.d_rnc = ptr(d_vals), ptr(d_fused), ptr(d_rnc)
        g.d_text_hidden, g.d_cross_text = ptr(d_text_hidden), ptr(d_cross_text)
        g.grads = ptr(grads)
        check(lib.sdumc_net_backward(C.byref(self.dims), C.byref(self.io), C.byref(g), _lib.current_stream()),
              "sdumc_net_backward")
        return grads


class _OptStateMixin:
    """Optimiser / RNG state of one training run: .adam_m .adam_v [live], .hyper = device {lr, step count t, lr/(1-b1^t),
    sqrt(1-b2^t)} (the last two are recomputed from t by every step), .rng = device Philox {seed, call}."""

    def optimizer_state(self):
        """(exp_avg, exp_avg_sq, step) -- what checkpoint.adam_state_from_flat takes."""
        return self.adam_m, self.adam_v, int(round(float(self.hyper[1].item())))

    def load_optimizer_state(self, adam_m, adam_v, step):
        """Resume (main_frame_val_text_missing.py:375 saves 'optimizer'; checkpoint.flat_from_adam_state gives the flat
        moments): installs the Adam moments, the step count that drives the bias correction, and the dropout call
        counter (two forward calls per step), so that the next step continues the interrupted run exactly."""
        if adam_m.numel() != self.adam_m.numel() or adam_v.numel() != self.adam_v.numel():
            raise _lib.SdumcError("load_optimizer_state: moment buffers must have layout.live elements")
        step = int(step)
        self.adam_m.copy_(adam_m.reshape(-1).to(self.adam_m.device))
        self.adam_v.copy_(adam_v.reshape(-1).to(self.adam_v.device))
        self.hyper[1] = float(step)
        self.rng.set_call(2 * step)


class TrainStep(_OptStateMixin):
    """The fused two-stream self-distillation step (main :119-150) on one GPU:
    forward(both streams) -> 6 losses -> backward -> Adam, ~150 launches on one stream,
    optionally captured into a hipGraph (torch.cuda.CUDAGraph) and replayed."""

    def __init__(self, flat_params, B, T, dims, weights=DEFAULT_WEIGHTS, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-5, seed=0, train=True, sample0=0, bf16=False, share=None, arena=None, ctx=None, planes=None,
                 bits_next=True):
        """share: an object with .params .rng .adam_m .adam_v .hyper .losses (another TrainStep over the SAME flat_params, or
        FusedTrainer's run state) whose optimiser state this step uses instead of allocating its own -- steps of different
        (B, T) shapes then continue one training run.
        arena: a _StepArena sized for the largest batch of the run: workspace, input and output buffers are views into it
        instead of fresh allocations (a C2 step owns ~1.2 GB of workspace: one arena per run, not one per batch shape); the arena
        decides about planes and owns the keep-bits buffer (its shapes change from step to step: use_set / launch(next_step=)).
        planes=True: the batch installed by set_batch is RESIDENT (run many times): its bf16 planes are split once, there (planes_wanted)."""
        Ta, Tt, Tv, T4 = T
        self.layout = ParamLayout.get(dims[0], dims[1], dims[2])
        dev = flat_params.device
        _require_cuda(flat_params)
        self.params = flat_params
        self.dims = make_dims(B, 2, Ta, Tv, (Tt, T4), dims, train, sample0, bf16=bf16)
        nbytes = lib.sdumc_step_workspace_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise _lib.SdumcError("invalid step dimensions")
        if share is not None and share.params.data_ptr() != flat_params.data_ptr():
            raise _lib.SdumcError("share: both steps must update the same flat parameter buffer")
        self.rng = share.rng if share is not None else RngState(seed, dev)
        V = 2 * B
        self.B, self.V, self.T = B, V, (Ta, Tt, Tv, T4)
        self._fdims = (dims[0], dims[1], dims[2], dims[1])
        fdt = torch.bfloat16 if self.dims.bf16 == 2 else torch.float32      # dtype the features are held in
        self.feature_dtype = fdt
        self._arena = arena
        self._lengths = None      # key-padding extension off (set_lengths / use_lengths)
        self._use_planes = False
        self._borrowed = None     # the caller's tensors of a zero-copy hand-over (use_batch)
        self._set = 0
        io = _lib.NetIO()
        self.io = io
        cfg = _lib.StepCfg()
        self.cfg = cfg
        self._planes = None
        self._bits_next = None
        if arena is not None:
            if arena.feature_dtype != fdt:
                raise _lib.SdumcError("arena and step disagree on the feature dtype")
            if not arena.fits(B, T, nbytes):
                raise _lib.SdumcError("arena too small for this batch shape")
            self.workspace = arena.workspace
            nbytes = arena.workspace.numel()
            self.vals = arena.outs[0][:V].view(V, 1)
            self.fused = arena.outs[1][:V * H].view(V, H)
            self.rnc = arena.outs[2][:V * RNC_DIM].view(V, RNC_DIM)
            self.text_hidden = arena.outs[3][:V * D].view(V, D)
            self.cross_text = arena.outs[4][:V * NQ * H].view(V, NQ, H)
            self._views = {}
            self.use_set(0)
            if arena.bits is not None and train:
                io.bits_next, io.bits_next_bytes = ptr(arena.bits), arena.bits.numel()
        else:
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self.audio = torch.empty(B, Ta, dims[0], device=dev, dtype=fdt)
            self.text = torch.empty(B, Tt, dims[1], device=dev, dtype=fdt)
            self.video = torch.empty(B, Tv, dims[2], device=dev, dtype=fdt)
            self.feat4 = torch.empty(B, T4, dims[1], device=dev, dtype=fdt)
            self.labels = torch.empty(B, device=dev)
            self.vals = torch.empty(V, 1, device=dev)
            self.fused = torch.empty(V, H, device=dev)
            self.rnc = torch.empty(V, RNC_DIM, device=dev)
            self.text_hidden = torch.empty(V, D, device=dev)
            self.cross_text = torch.empty(V, NQ, H, device=dev)
            io.audio, io.video = ptr(self.audio), ptr(self.video)
            io.text[0], io.text[1] = ptr(self.text), ptr(self.feat4)
            cfg.labels = ptr(self.labels)
            # fp32 storage, a RESIDENT batch (planes=True): bf16-plane copies of the four feature tensors, split once by set_batch --
            # 1.5x the features' bytes on top of them
            if planes_wanted(planes, dims, bf16):
                self._use_planes = True
                self._planes = [torch.empty(t.shape[0] * t.shape[1], 6 * t.shape[2], dtype=torch.uint8, device=dev)
                                for t in (self.audio, self.text, self.video, self.feat4)]
                io.audio_p3, io.text_p3[0], io.video_p3, io.text_p3[1] = (ptr(t) for t in self._planes)
            # fp32 train steps: a buffer of its own for the NEXT step's keep-bits (generated in this step's idle middle, found at the next
            # step's head under its {seed, call, shape} tag; bit-identical masks) -- not in the workspace: it must survive whatever else
            # runs between two steps of this shape
            nb = lib.sdumc_net_bits_next_bytes(C.byref(self.dims)) if bits_next else 0
            self._bits_next = torch.zeros(nb, dtype=torch.uint8, device=dev) if nb else None
            if self._bits_next is not None:
                io.bits_next = ptr(self._bits_next)
        if share is not None:
            self.adam_m, self.adam_v, self.hyper, self.losses = share.adam_m, share.adam_v, share.hyper, share.losses
        else:
            self.adam_m = torch.zeros(self.layout.live, device=dev)
            self.adam_v = torch.zeros(self.layout.live, device=dev)
            self.hyper = torch.tensor([lr, 0.0, 0.0, 0.0], device=dev)
            self.losses = torch.zeros(8, device=dev)
        io.params, io.rng_state = ptr(flat_params), ptr(self.rng.t)
        io.workspace, io.workspace_bytes = ptr(self.workspace), nbytes
        io.vals, io.fused, io.rnc = ptr(self.vals), ptr(self.fused), ptr(self.rnc)
        io.text_hidden, io.cross_text = ptr(self.text_hidden), ptr(self.cross_text)
        self._ctx = ctx       # ExecContext or None (= the device's default lanes)
        io.ctx = ctx.handle if ctx is not None else None
        for i, w in enumerate(weights):
            cfg.weights[i] = w
        cfg.temperature = 2.0
        cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay = betas[0], betas[1], eps, weight_decay
        cfg.adam_m, cfg.adam_v = ptr(self.adam_m), ptr(self.adam_v)
        cfg.hyper, cfg.losses = ptr(self.hyper), ptr(self.losses)
        self.graph = None
        goff = lib.sdumc_step_grads_offset(C.byref(self.dims))      # the same for every shape: the bucket leads the workspace
        self.grads = self.workspace[goff:goff + 4 * self.layout.live].view(torch.float32)
        if arena is None:
            self.grads.zero_()      # alignment padding between tensors is never written by the kernels (an arena zeroes it once)

    def use_set(self, k, planes=True):
        """Arena steps: read the batch held in input set `k` of the arena (FusedTrainer alternates two sets -- the next batch is
        assembled in the other one while this step runs; bench.py rotates K resident batches).  planes=False: this batch has no
        planes in the set (fresh tensors handed to step(): splitting them for one use costs more than it saves).  Pointers only."""
        a = self._arena
        if a is None:
            raise _lib.SdumcError("use_set: this step owns its input buffers (no arena)")
        st = a.sets[k]
        v = self._views.get(k)
        if v is None:
            B, T, fd = self.B, self.T, self._fdims
            v = tuple(st.inputs[i][:B * T[i] * fd[i]].view(B, T[i], fd[i]) for i in range(4)) + (st.labels[:B],)
            self._views[k] = v
        self.audio, self.text, self.video, self.feat4, self.labels = v
        io = self.io
        io.audio, io.text[0], io.video, io.text[1] = (ptr(t) for t in st.inputs)
        for i in range(4):
            io.row_map[i] = None
            io.store_rows[i] = 0
        self._use_planes = bool(planes) and st.planes is not None
        if self._use_planes:
            io.audio_p3, io.text_p3[0], io.video_p3, io.text_p3[1] = (ptr(t) for t in st.planes)
        else:
            io.audio_p3 = io.video_p3 = io.text_p3[0] = io.text_p3[1] = None
        self.cfg.labels = ptr(st.labels)
        self._set = k
        self._borrowed = None
        return self

    def use_store(self, store, k):
        """Arena steps in fp32 storage: read the batch IN PLACE from a DeviceFeatureStore(planes=True) through the row maps held in
        input set `k` (written by the store's gather_desc(maps_out=...) launch): the step's feature pointers are the store's packed
        tensors, no padded copy of the batch exists.  Labels / lengths come from the set as with use_set."""
        a = self._arena
        hf = self.dims.bf16 == 2
        if a is None or self.dims.bf16 == 1 or (not hf and store.packed_p3 is None) or store.packed['audio'].dtype != self.feature_dtype:
            raise _lib.SdumcError("use_store: an arena step in fp32 storage with a store that holds planes, or in bf16 storage with a bf16 store")
        st = a.sets[k]
        io = self.io
        pk = store.packed
        io.audio, io.text[0], io.video, io.text[1] = ptr(pk['audio']), ptr(pk['text']), ptr(pk['video']), ptr(pk['feat4'])
        if not hf:
            p3 = store.packed_p3
            io.audio_p3, io.text_p3[0], io.video_p3, io.text_p3[1] = ptr(p3['audio']), ptr(p3['text']), ptr(p3['video']), ptr(p3['feat4'])
        maps = st.ensure_maps()
        for i, m in enumerate(('audio', 'text', 'video', 'feat4')):
            io.row_map[i] = ptr(maps[i])
            io.store_rows[i] = int(pk[m].shape[0])
        self.labels = st.labels[:self.B]
        self.cfg.labels = ptr(st.labels)
        self._use_planes, self._set = not hf, k
        self._borrowed = None
        return self

    def _point_lengths(self, tensors):
        for i in range(4):
            self.io.lengths[i] = ptr(tensors[i]) if tensors is not None else None

    def set_batch(self, audio, text, video, feat4, labels):
        """Copies one batch into the step's resident input buffers (shapes are fixed per TrainStep); in bf16-storage mode the
        buffers are bf16 and fp32 inputs are rounded by the copy.  With planes (a resident batch) the bf16 planes are split here."""
        self._restore_own_inputs()
        self.audio.copy_(audio, non_blocking=True)
        self.text.copy_(text, non_blocking=True)
        self.video.copy_(video, non_blocking=True)
        self.feat4.copy_(feat4, non_blocking=True)
        self.labels.copy_(labels.reshape(-1), non_blocking=True)
        planes = None
        if self._use_planes:
            planes = self._planes if self._arena is None else self._arena.sets[self._set].planes
        if planes is not None:
            for src, dst in zip((self.audio, self.text, self.video, self.feat4), planes):
                p3_split_into(src, dst)

    def use_batch(self, audio, text, video, feat4, labels):
        """Zero-copy hand-over of one batch: the step reads the caller's device tensors where they are (no 224 MB device copy per
        batch: what `for data in loader: step.use_batch(*unpack(data, 'cuda')[:5]); step.run()` saves over set_batch).  The tensors
        must have the step's shapes and feature dtype, be contiguous and 16-byte aligned, and stay alive and unchanged until the step
        has run (the step keeps references until the next hand-over); labels [B] fp32.  Returns False -- and copies, as set_batch --
        when a tensor does not qualify (fp32 features handed to a bf16-storage step are rounded by the copy).  Not for captured
        steps (a graph replays the addresses it was recorded with)."""
        if self.graph is not None:
            raise _lib.SdumcError("use_batch: the captured hipGraph reads the step's own buffers (set_batch)")
        ts = (audio, text, video, feat4)
        ok = labels.is_cuda and labels.dtype == torch.float32 and labels.is_contiguous() and labels.numel() == self.B
        for t, T, d in zip(ts, self.T, self._fdims):
            ok = ok and t.is_cuda and t.dtype == self.feature_dtype and t.is_contiguous() and tuple(t.shape) == (self.B, T, d) \
                and t.data_ptr() % 16 == 0
        if not ok:
            self.set_batch(audio, text, video, feat4, labels)
            return False
        io = self.io
        io.audio, io.text[0], io.video, io.text[1] = ptr(audio), ptr(text), ptr(video), ptr(feat4)
        io.audio_p3 = io.video_p3 = io.text_p3[0] = io.text_p3[1] = None      # (a fresh batch: no planes to split for one use)
        for i in range(4):
            io.row_map[i] = None
            io.store_rows[i] = 0
        self.cfg.labels = ptr(labels.reshape(-1))
        self._borrowed = (audio, text, video, feat4, labels)
        self._use_planes = False
        return True

    def _restore_own_inputs(self):
        """after use_batch: point the step back at its own input buffers (set_batch writes those)"""
        if self._borrowed is None:
            return
        self._borrowed = None
        if self._arena is not None:
            self.use_set(self._set)
            return
        io = self.io
        io.audio, io.video = ptr(self.audio), ptr(self.video)
        io.text[0], io.text[1] = ptr(self.text), ptr(self.feat4)
        self.cfg.labels = ptr(self.labels)
        if self._planes is not None:
            io.audio_p3, io.text_p3[0], io.video_p3, io.text_p3[1] = (ptr(t) for t in self._planes)
            self._use_planes = True

    def set_lengths(self, lengths):
        """Key-padding extension: (audio, text, video, feat4) valid frame counts of the current batch, or None to go back
        to the reference's behaviour (padded frames take part in the softmax)."""
        new = _lengths_arg(lengths, self.B, 4, self.params.device)
        if self.graph is not None and (new is None) != (self._lengths is None):
            raise _lib.SdumcError("the captured hipGraph was recorded with the key-padding extension "
                                  + ("on" if self._lengths is not None else "off") + ": switch it before capture()")
        if new is None:
            self._lengths = None
            self._point_lengths(None)
            return
        if self._lengths is None:
            self._lengths = [torch.empty(self.B, dtype=torch.int32, device=self.params.device) for _ in range(4)]
        self._point_lengths(self._lengths)
        for dst, src in zip(self._lengths, new):
            dst.copy_(src, non_blocking=True)      # resident buffers: a captured graph keeps reading the same addresses

    def use_lengths(self, tensors):
        """Arena steps: the key-padding lengths as four device int32 buffers the batch's assembly already filled (or None = off)."""
        self._lengths = list(tensors) if tensors is not None else None
        self._point_lengths(self._lengths)

    def set_lr(self, lr):
        self.hyper[0] = lr

    def launch(self, stream=None, next_step=None, prefetch=None, pregen=True):
        """Enqueues the step.  next_step: the TrainStep that runs NEXT in this arena (its dims decide how this step's middle lays out
        the next keep-bits set); prefetch: the sdumc_gather_batch descriptor of the next batch, issued by the step beside its middle
        and backward; pregen=False: no keep-bits for a next step this time (the caller does not expect a step of a known shape to follow)."""
        st = _lib.current_stream() if stream is None else stream
        io = self.io
        a = self._arena
        shared = a is not None and a.bits is not None and io.bits_next
        if shared:
            io.bits_phase = a.bits_phase
        saved = io.bits_next
        if not pregen and self.graph is None:
            io.bits_next = None
        io.bits_next_dims = C.addressof(next_step.dims) if next_step is not None else None
        io.prefetch = C.addressof(prefetch) if prefetch is not None else None
        io.prefetch_workgroups = a.prefetch_workgroups if a is not None else 0
        try:
            check(lib.sdumc_train_step(C.byref(self.dims), C.byref(io), C.byref(self.cfg), st), "sdumc_train_step")
        finally:
            io.bits_next, io.bits_next_dims, io.prefetch = saved, None, None
        if pregen or self.graph is not None:
            if shared:
                a.bits_phase ^= 1
            else:
                io.bits_phase ^= 1      # (the next step reads the keep-bits set this one filled in its middle: sdumc_net_io.bits_next)

    def capture(self):
        """Capture one step into a hipGraph; run() then replays it.  For embedding the step in a captured region, not for speed:
        the capture records a plain three-lane fork/join (no background lane, chain.hip instead of the clustered kernels) and
        measured slower than the eager four-lane launches, which are the production path."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        # the capture itself does not execute; state (params, Adam, rng) is untouched by it
        self.io.bits_next = None      # (a replayed graph cannot alternate the two keep-bits sets: every replay generates its own)
        with torch.cuda.graph(g, stream=s):
            self.launch()
        self.graph = g

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.launch()
        return self.losses


class _RunState(_OptStateMixin):
    """Optimiser state of one training run, shared by the per-shape TrainSteps of a FusedTrainer (and by the per-shape
    backends of trainer.DataParallelStep)."""

    def __init__(self, flat_params, live, lr, seed):
        dev = flat_params.device
        self.params = flat_params
        self.rng = RngState(seed, dev)
        self.adam_m = torch.zeros(live, device=dev)
        self.adam_v = torch.zeros(live, device=dev)
        self.hyper = torch.tensor([lr, 0.0, 0.0, 0.0], device=dev)
        self.losses = torch.zeros(8, device=dev)


class _InputSet:
    """One resident batch slot of an arena: the four feature buffers (flat, capacity-sized), their P3 planes (fp32 storage, planes on),
    the labels and the valid frame counts."""

    def __init__(self, n, planes_bytes, B, dtype, dev, rows):
        self._n, self._dtype, self._inputs = n, dtype, None      # (the padded copies' buffers exist only once something is copied)
        self.planes = [torch.empty(k, device=dev, dtype=torch.uint8) for k in planes_bytes] if planes_bytes is not None else None
        self.labels = torch.empty(B, device=dev)
        self.lengths = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
        self.maps = None      # row maps (int32 per frame): a batch read in place from a DeviceFeatureStore (ensure_maps)
        self._rows = rows

    @property
    def inputs(self):
        if self._inputs is None:
            self._inputs = [torch.empty(k, device=self.labels.device, dtype=self._dtype) for k in self._n]
        return self._inputs

    def ensure_maps(self):
        if self.maps is None:
            # (+ 64 entries: the bf16 weight-gradient kernel fetches map entries four at a time, past a batch's last row)
            self.maps = [torch.zeros(r + 64, dtype=torch.int32, device=self.labels.device) for r in self._rows]
        return self.maps


class _StepArena:
    """Device memory of one training run, sized once for its largest batch (B, T_audio, T_text, T_video, T_feat4):
    the step workspace (whose leading gradient bucket is zeroed here, once), `sets` input sets the batches are assembled in --
    DeviceFeatureStore gathers straight into them; two sets let the NEXT batch be assembled while a step runs --, the five outputs,
    and (fp32 train steps) the two keep-bits sets every shape of the run shares (their tags name the shape they were laid out for)."""

    def __init__(self, flat_params, B, T, dims, bf16=False, sets=1, planes=False, bits_next=True, prefetch_workgroups=0):
        dev = flat_params.device
        self.B, self.T, self.dims = int(B), tuple(int(t) for t in T), tuple(dims)
        d = make_dims(self.B, 2, self.T[0], self.T[2], (self.T[1], self.T[3]), dims, True, 0, bf16=bf16)
        nbytes = lib.sdumc_step_workspace_bytes(C.byref(d))
        if nbytes == 0:
            raise _lib.SdumcError("invalid arena dimensions")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        lay = ParamLayout.get(dims[0], dims[1], dims[2])
        goff = lib.sdumc_step_grads_offset(C.byref(d))
        self.workspace[goff:goff + 4 * lay.live].zero_()
        fd = (dims[0], dims[1], dims[2], dims[1])
        n = [self.B * self.T[i] * fd[i] for i in range(4)]
        self.feature_dtype = torch.bfloat16 if d.bf16 == 2 else torch.float32
        # planes: True = the input sets carry P3 planes from the start; None = as soon as a store with planes (or a resident batch)
        # asks for them (ensure_planes); False = never
        self._planes_ok = planes is not False and planes_wanted(True, dims, bf16)
        self._n = n
        rows = [self.B * self.T[i] for i in range(4)]
        self.sets = [_InputSet(n, None, self.B, self.feature_dtype, dev, rows) for _ in range(max(1, int(sets)))]
        if planes is True:
            self.ensure_planes()
        V = 2 * self.B
        self.outs = [torch.empty(V * k, device=dev) for k in (1, H, RNC_DIM, D, NQ * H)]
        nb = lib.sdumc_net_bits_next_bytes(C.byref(d)) if bits_next else 0
        self.bits = torch.zeros(nb, dtype=torch.uint8, device=dev) if nb else None
        self.bits_phase = 0
        self.prefetch_workgroups = int(prefetch_workgroups)

    def ensure_planes(self):
        """The plane buffers of every input set (1.5x the fp32 input bytes each), allocated on first need; False when the arena's
        mode has none (bf16 storage, widths that are not whole k-tiles, planes=False)."""
        if not self._planes_ok:
            return False
        for st in self.sets:
            if st.planes is None:
                st.planes = [torch.empty(6 * k, device=self.workspace.device, dtype=torch.uint8) for k in self._n]
        return True

    # (the single-set views older callers used)
    @property
    def inputs(self):
        return self.sets[0].inputs

    @property
    def labels(self):
        return self.sets[0].labels

    @property
    def lengths(self):
        return self.sets[0].lengths

    def fits(self, B, T, nbytes):
        return B <= self.B and all(t <= c for t, c in zip(T, self.T)) and nbytes <= self.workspace.numel()


StepArena = _StepArena      # (public name: bench.py and callers that rotate resident batches build one directly)


class FusedTrainer:
    """The fused step for the reference's REAL batches, whose (B, T_audio, T_text, T_video, T_feat4) change from batch to
    batch (every modality is padded to its batch maximum, read_data.py:223-248; the last batch of an epoch is short):
    one TrainStep per shape, created on first use, all sharing one optimiser state, so `step()` over a data loader is one
    continuous run of main_frame_val_text_missing.py:119-150.  At most `max_cached` shapes keep their workspace (least
    recently used first out); a shape seen again after eviction is simply rebuilt.

    With capacity= and a DeviceFeatureStore, `run_epoch` is the replacement for the reference's loop over its DataLoader
    (main :89-109 over feat_data.py:232-253): the epoch's index vectors are uploaded once, every batch is assembled on the device
    straight into one of the arena's two input sets -- by the PREVIOUS step, beside its latency-bound middle and its backward
    (sdumc_net_io.prefetch) -- and every step tells the engine the next batch's shape, so that the next keep-bits are laid out for it."""

    def __init__(self, flat_params, dims, max_cached=8, lr=1e-4, seed=0, capacity=None, planes=None, sets=2, prefetch_workgroups=0,
                 inplace=True, **step_kwargs):
        """capacity = (B_max, (T_audio, T_text, T_video, T_feat4) maxima) of the run: ONE arena then backs every batch shape
        (no per-shape workspace, the per-shape step is a few ctypes structs and tensor views: cache as many as you like) and
        `step_from_store` / `run_epoch` assemble batches straight into it.  Without it every cached shape owns its workspace.
        planes (arena only): None (default) = follow the store -- batches assembled from a DeviceFeatureStore(planes=True) bring their
        P3 planes along (the same gather as the fp32 rows) and the step's frame projections read them; False = never.
        inplace (default True): with such a store the batches are not gathered at all -- the step reads the store's packed tensors
        through per-batch row maps (4 bytes per frame; sdumc_net_io.row_map)."""
        _require_cuda(flat_params)
        self.params, self.dims, self.max_cached = flat_params, tuple(dims), max(1, int(max_cached))
        self.kw = dict(step_kwargs, lr=lr, seed=seed)
        lay = ParamLayout.get(dims[0], dims[1], dims[2])
        self.state = _RunState(flat_params, lay.live, lr, seed)     # (params, rng, adam_m, adam_v, hyper, losses)
        self._steps = {}          # shape -> TrainStep, insertion order = recency
        self.arena = None
        self._arena_kw = dict(bf16=step_kwargs.get("bf16", False), sets=sets, planes=planes,
                              bits_next=step_kwargs.get("bits_next", True), prefetch_workgroups=prefetch_workgroups)
        self._last_shape = None
        # inplace (fp32 storage, a store with planes): batches are read from the store through row maps, no padded copy (use_store);
        # False: every batch is gathered into the arena's input sets (fp32 rows + plane rows)
        self.inplace = bool(inplace)
        if capacity is not None:
            self.arena = _StepArena(flat_params, capacity[0], capacity[1], dims, **self._arena_kw)
            self.max_cached = max(self.max_cached, 4096)
        else:
            # per-shape steps own their buffers and see a fresh batch per step: no planes to re-split, and no keep-bits set kept per
            # shape (the next step is usually another shape with another buffer: its tag would never match)
            self.kw.update(planes=False, bits_next=False)

    def _get(self, B, T):
        key = (B,) + tuple(T)
        ts = self._steps.pop(key, None)
        if ts is None:
            while len(self._steps) >= self.max_cached:
                del self._steps[next(iter(self._steps))]
            if self.arena is not None and not (B <= self.arena.B and all(t <= c for t, c in zip(T, self.arena.T))):
                # a batch beyond the declared capacity: grow the arena (every cached step pointed into the old one)
                cap_T = tuple(max(t, c) for t, c in zip(T, self.arena.T))
                self.arena = _StepArena(self.params, max(B, self.arena.B), cap_T, self.dims, **self._arena_kw)
                self._steps.clear()
            kw = {k: v for k, v in self.kw.items() if not (self.arena is not None and k in ("planes", "bits_next"))}
            ts = TrainStep(self.params, B, T, self.dims, share=self.state, arena=self.arena, **kw)
        self._steps[key] = ts
        return ts

    def set_lr(self, lr):
        self.state.hyper[0] = lr

    def load_optimizer_state(self, adam_m, adam_v, step):
        self.state.load_optimizer_state(adam_m, adam_v, step)

    def optimizer_state(self):
        return self.state.optimizer_state()

    def _launch(self, ts, next_step=None, prefetch=None):
        # keep-bits for a next step only when its shape is known (run_epoch) or has been repeating (a static-shape loader)
        key = (ts.B,) + ts.T
        pregen = next_step is not None or key == self._last_shape
        self._last_shape = key
        ts.launch(next_step=next_step, prefetch=prefetch, pregen=pregen)
        return ts.losses

    def _store_planes(self, store):
        return store.packed_p3 is not None and self.arena.ensure_planes()

    def _in_place(self, store):
        """fp32 storage and a store with planes, or bf16 storage and a bf16 store (widths in whole 128-element tiles): batches are not
        copied at all -- the step reads the store's packed tensors through row maps."""
        if not self.inplace:
            return False
        if self.arena.feature_dtype == torch.bfloat16:
            return store.packed['audio'].dtype == torch.bfloat16 and all(int(d) % 128 == 0 for d in self.dims[:3])
        return store.packed_p3 is not None and self.arena._planes_ok

    def _gather_desc(self, store, idx_ptr, ts, k, key_padding, planes, inplace=False):
        st = self.arena.sets[k]
        if inplace:
            return store.gather_desc(idx_ptr, ts.B, ts.T, None, st.labels, st.lengths if key_padding else None, maps_out=st.ensure_maps())
        return store.gather_desc(idx_ptr, ts.B, ts.T, st.inputs, st.labels, st.lengths if key_padding else None, st.planes if planes else None)

    def step_from_store(self, store, indices, key_padding=False):
        """One optimisation step on the batch `indices` of a data.DeviceFeatureStore: the padded batch is assembled by the
        gather/pad kernel DIRECTLY in the step's input buffers (no intermediate batch tensors, no 224 MB device copy);
        key_padding=True also hands the valid frame counts to the kernels (extension, default off = the reference).
        (One batch at a time: the assembly runs in front of the step.  run_epoch overlaps it with the previous step.)"""
        if self.arena is None:
            raise _lib.SdumcError("step_from_store needs FusedTrainer(capacity=...)")
        B, T = store.batch_shape(indices)
        ts = self._get(B, T)
        st = self.arena.sets[0]
        if self._in_place(store):
            idx_d = store._checked(indices).to(store.device, non_blocking=True)
            g = self._gather_desc(store, idx_d.data_ptr(), ts, 0, key_padding, True, inplace=True)
            check(lib.sdumc_gather_batch(C.byref(g), 0, _lib.current_stream()), "sdumc_gather_batch")
            self._keep_idx = idx_d
            ts.use_store(store, 0)
        else:
            planes = self._store_planes(store)
            ts.use_set(0, planes=planes)
            store.batch_into(indices, (ts.audio, ts.text, ts.video, ts.feat4), ts.labels, st.lengths if key_padding else None,
                             st.planes if planes else None)
        ts.use_lengths(st.lengths if key_padding else None)
        return self._launch(ts)

    def run_epoch(self, store, batches, key_padding=False, on_step=None):
        """Every batch of `batches` (index vectors into `store`, or a data.EpochPlan), in order: main :89-150's loop.  The first batch
        is assembled in front of the first step; from then on step i assembles batch i + 1 in the other input set while it runs.
        on_step(i, losses) is called after step i has been ENQUEUED (losses = the device vector the step will write: clone it to keep
        it).  Returns the number of steps."""
        if self.arena is None or len(self.arena.sets) < 2:
            raise _lib.SdumcError("run_epoch needs FusedTrainer(capacity=..., sets=2)")
        from .data import EpochPlan
        plan = batches if isinstance(batches, EpochPlan) else store.plan_epoch(batches)
        n = len(plan)
        # the arena must hold the epoch's largest batch BEFORE the first descriptor is built (growing it invalidates every cached step);
        # the per-shape step objects themselves are made one step ahead of the GPU, inside the loop: the host runs ahead of the device
        # anyway, so a shape seen for the first time costs no device time
        Bm = max(B for B, _ in plan.shapes)
        Tm = tuple(max(T[i] for _, T in plan.shapes) for i in range(4))
        if not (Bm <= self.arena.B and all(t <= c for t, c in zip(Tm, self.arena.T))):
            self._get(max(Bm, self.arena.B), tuple(max(t, c) for t, c in zip(Tm, self.arena.T)))
        inplace = self._in_place(store)
        planes = inplace or self._store_planes(store)
        nxt = self._get(*plan.shapes[0])
        g0 = self._gather_desc(store, plan.idx_ptr(0), nxt, 0, key_padding, planes, inplace)
        check(lib.sdumc_gather_batch(C.byref(g0), 0, _lib.current_stream()), "sdumc_gather_batch")
        for i in range(n):
            ts = nxt.use_store(store, i & 1) if inplace else nxt.use_set(i & 1, planes=planes)
            ts.use_lengths(self.arena.sets[i & 1].lengths if key_padding else None)
            nxt = self._get(*plan.shapes[i + 1]) if i + 1 < n else None
            pf = self._gather_desc(store, plan.idx_ptr(i + 1), nxt, (i + 1) & 1, key_padding, planes, inplace) if nxt is not None else None
            losses = self._launch(ts, next_step=nxt, prefetch=pf)
            if on_step is not None:
                on_step(i, losses)
        self._keep_plan = plan      # (the index tensor must outlive the enqueued gathers)
        return n

    def step(self, audio, text, video, feat4, labels, lengths=None):
        """One optimisation step on one batch of any shape; returns the device loss vector
        [total, mse_full, mse_missing, rmse_text, rmse_query, rmse_fused, rnc, 0]."""
        ts = self._get(audio.shape[0], (audio.shape[1], text.shape[1], video.shape[1], feat4.shape[1]))
        if self.arena is not None:
            ts.use_set(0, planes=False)      # (fresh tensors: a split for one use costs more than the planes save)
        # the caller's device tensors are read where they are when they qualify (no 224 MB copy per batch); else copied
        ts.use_batch(audio, text, video, feat4, labels.reshape(-1) if torch.is_tensor(labels) else labels)
        ts.set_lengths(lengths)
        return self._launch(ts)
