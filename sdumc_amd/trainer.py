"""Data-parallel two-stream self-distillation step (the reference has no DP: SURVEY §2.2, §8e).

One process per GPU, plain batch sharding, weights replicated.  Per step:
  1. local forward of both streams                                   (HIP, sdumc_net_forward)
  2. exactness exchange so that DP(N x B) == single process(N*B), ONE all-gather of a per-rank record holding
       - the three RMSE sums of squared differences (RMSELoss is the sqrt of a GLOBAL mean,
         toolkit/utils/loss.py:37-51), added in rank order on every rank
       - the RnC embeddings and labels (RnCLoss uses in-batch negatives over n = 2*B_global,
         loss.py:271-315); every rank evaluates the full loss and keeps the gradient of its own rows
  3. loss gradients w.r.t. the local outputs, backward                (HIP)
  4. all-reduce (sum) of the flat gradient bucket over RCCL/xGMI (15.4 MB fp32): one call after the backward (default),
     or, with SDUMC_DP_OVERLAP=1, in two slices -- the utterance-level layers' slice asynchronously as soon as backward
     phase 0 has produced it (it then overlaps the frame-level backward, 0.9 ms of GEMMs), the frame-level slice at the end
  5. fused Adam on the flat bucket                                    (HIP)
Dropout masks are keyed by the GLOBAL sample index, so results do not depend on N.

The compute backend is injectable: the product default is HipBackend (no fallback); tests inject
a CPU backend to check the collective algebra under gloo.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class HipBackend:
    """Per-rank compute on one MI355X through the C ABI."""

    def __init__(self, flat_params, B, T, dims, weights, lr, betas, eps, weight_decay, seed, sample0, B_global,
                 bf16=False, share=None, planes=False):
        """share: the run state (engine._RunState: rng, Adam moments, hyper, losses, gradient bucket) of the training run
        this backend belongs to.  planes=True: the batches installed by set_batch are RESIDENT (each runs many steps): their bf16
        planes are split once per set_batch (engine.planes_wanted); default off -- a fresh batch per step would pay the split for one use.
        The reference's loader pads every batch to its own max T and ends an epoch on a short
        batch (read_data.py:223-248), so a run needs one backend per (B, T) shape; they must all continue ONE optimiser
        state, step count and dropout call counter (DataParallelStep keeps them in an LRU and passes the same `share`)."""
        from . import _lib, engine
        self._lib, self._engine = _lib, engine
        lib = _lib.lib
        dev = flat_params.device
        Ta, Tt, Tv, T4 = T
        self.B, self.B_global = B, B_global
        self.layout = engine.ParamLayout.get(dims[0], dims[1], dims[2])
        self.params = flat_params
        fdt = torch.bfloat16 if engine.bf16_mode(bf16, dims) == 2 else torch.float32     # bf16-storage mode holds bf16 features
        self.audio = torch.empty(B, Ta, dims[0], device=dev, dtype=fdt)
        self.text = torch.empty(B, Tt, dims[1], device=dev, dtype=fdt)
        self.video = torch.empty(B, Tv, dims[2], device=dev, dtype=fdt)
        self.feat4 = torch.empty(B, T4, dims[1], device=dev, dtype=fdt)
        self.labels = torch.empty(B, device=dev)
        st = share if share is not None else engine._RunState(flat_params, self.layout.live, lr, seed)
        if st.params.data_ptr() != flat_params.data_ptr():
            raise _lib.SdumcError("share: the run state belongs to another parameter buffer")
        self.state = st
        self.rng = st.rng
        self.call = engine.NetCall(flat_params, self.audio, [self.text, self.feat4], self.video, True, self.rng,
                                   sample0=sample0, bf16=bf16, planes=planes, bits_next=True)
        V = 2 * B
        self.d_vals = torch.empty(V, 1, device=dev)
        self.d_fused = torch.empty(V, engine.H, device=dev)
        self.d_rnc = torch.empty(V, engine.RNC_DIM, device=dev)
        self.d_text_hidden = torch.empty(V, engine.D, device=dev)
        self.d_cross_text = torch.empty(V, engine.NQ, engine.H, device=dev)
        if getattr(st, "grads", None) is None:
            # the all-reduce bucket: one per run, not per shape; 4 trailing floats carry this rank's clustered-kernel error
            # word through the same collective (err_flag / err_merge)
            st.bucket = torch.zeros(self.layout.live + 4, device=dev)
            st.grads = st.bucket[:self.layout.live]
        self.grads, self.bucket = st.grads, st.bucket
        self.adam_m, self.adam_v, self.hyper, self.losses = st.adam_m, st.adam_v, st.hyper, st.losses
        self.ssd = torch.zeros(4, device=dev)
        nb = lib.sdumc_loss_workspace_bytes(C.byref(self.call.dims), B_global)
        self.loss_ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        cfg = _lib.StepCfg()
        for i, w in enumerate(weights):
            cfg.weights[i] = w
        cfg.temperature = 2.0
        cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay = betas[0], betas[1], eps, weight_decay
        cfg.labels, cfg.adam_m, cfg.adam_v = _lib.ptr(self.labels), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v)
        cfg.hyper, cfg.losses = _lib.ptr(self.hyper), _lib.ptr(self.losses)
        cfg.B_global = B_global
        self.cfg = cfg
        g = _lib.NetGrads()
        g.d_vals, g.d_fused, g.d_rnc = _lib.ptr(self.d_vals), _lib.ptr(self.d_fused), _lib.ptr(self.d_rnc)
        g.d_text_hidden, g.d_cross_text = _lib.ptr(self.d_text_hidden), _lib.ptr(self.d_cross_text)
        g.grads = _lib.ptr(self.grads)
        self.g = g
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self._record = self._gathered = None

    def set_batch(self, audio, text, video, feat4, labels):
        self.audio.copy_(audio, non_blocking=True)
        self.text.copy_(text, non_blocking=True)
        self.video.copy_(video, non_blocking=True)
        self.feat4.copy_(feat4, non_blocking=True)
        self.labels.copy_(labels.reshape(-1), non_blocking=True)
        self.call.refresh_planes()      # (fp32 storage: the bf16-plane copies the frame projections read)

    def set_lengths(self, lengths):
        """Key-padding extension (default None = the reference's behaviour): (audio, text, video, feat4) valid frame counts."""
        self.call.set_lengths(lengths)

    def load_optimizer_state(self, adam_m, adam_v, step):
        self.state.load_optimizer_state(adam_m, adam_v, step)

    def forward(self):
        self.call.forward()
        return self.call.rnc                     # [2B, 64], stream-major

    def local_ssd(self):
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_loss_ssd(C.byref(self.call.dims), C.byref(self.call.io), _lib.ptr(self.ssd),
                                      _lib.ptr(self.loss_ws), self.loss_ws.numel(), _lib.current_stream()),
                   "sdumc_loss_ssd")
        return self.ssd[:3]

    def dp_pack(self):
        """This rank's exchange record [rnc features | labels | local sums of squares] in one launch."""
        lib, _lib = self._lib.lib, self._lib
        B, rd = self.B, self._engine.RNC_DIM
        if self._record is None:
            dev = self.params.device
            self._record = torch.empty(2 * B * rd + B + 3, device=dev)
            self._record_ws = torch.zeros(lib.sdumc_dp_record_workspace_bytes(B), dtype=torch.uint8, device=dev)
        c = self.call
        _lib.check(lib.sdumc_dp_record(B, rd, _lib.ptr(c.text_hidden), _lib.ptr(c.cross_text), _lib.ptr(c.fused),
                                       _lib.ptr(c.rnc), _lib.ptr(self.labels), _lib.ptr(self._record),
                                       _lib.ptr(self._record_ws), _lib.current_stream()), "sdumc_dp_record")
        return self._record

    def dp_unpack(self, records, W):
        """gathered records [W, n] -> (ssd[3], feats [2*W*B, 64], labels2 [2*W*B]) in the layout loss_backward takes."""
        lib, _lib = self._lib.lib, self._lib
        B, rd = self.B, self._engine.RNC_DIM
        if self._gathered is None or self._gathered[1].shape[0] != 2 * W * B:
            dev = self.params.device
            self._gathered = (torch.empty(4, device=dev), torch.empty(2 * W * B, rd, device=dev),
                              torch.empty(2 * W * B, device=dev))
        ssd, feats, labels2 = self._gathered
        _lib.check(lib.sdumc_dp_unpack(_lib.ptr(records), W, B, rd, _lib.ptr(feats), _lib.ptr(labels2), _lib.ptr(ssd),
                                       _lib.current_stream()), "sdumc_dp_unpack")
        return ssd, feats, labels2

    def loss_backward(self, ssd_global=None, feats_global=None, labels_global=None, row0=(0, 0)):
        lib, _lib = self._lib.lib, self._lib
        self._keep = (ssd_global, feats_global, labels_global)
        self.cfg.ssd_global = _lib.ptr(ssd_global)
        self.cfg.rnc_feats_global = _lib.ptr(feats_global)
        self.cfg.rnc_labels_global = _lib.ptr(labels_global)
        self.cfg.rnc_row0[0], self.cfg.rnc_row0[1] = row0
        _lib.check(lib.sdumc_loss_backward(C.byref(self.call.dims), C.byref(self.call.io), C.byref(self.cfg),
                                           C.byref(self.g), _lib.ptr(self.loss_ws), self.loss_ws.numel(),
                                           _lib.current_stream()), "sdumc_loss_backward")
        return self.losses

    def backward(self):
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_net_backward(C.byref(self.call.dims), C.byref(self.call.io), C.byref(self.g),
                                          _lib.current_stream()), "sdumc_net_backward")
        return self.grads

    def backward_phase(self, phase):
        """phase 0: utterance-level layers -> grads[:layout.early] final; phase 1: frame-level layers -> the rest."""
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_net_backward_phase(C.byref(self.call.dims), C.byref(self.call.io), C.byref(self.g), phase,
                                                _lib.current_stream()), "sdumc_net_backward_phase")
        return self.grads[:self.layout.early] if phase == 0 else self.grads[self.layout.early:]

    def err_flag(self):
        """bucket[live] = 1.0 if a clustered utterance-level kernel of this rank ran into its spin cap (its gradients are
        garbage), else 0.0 -- written on the stream, no host sync; the gradient all-reduce then carries it to every rank."""
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_chain_cluster_error_flag(_lib.ptr(self.bucket[self.layout.live:]), _lib.current_stream()),
                   "sdumc_chain_cluster_error_flag")

    def err_merge(self):
        """after the all-reduce: any rank failed -> this rank's error word is set too: its Adam applies nothing, it raises too"""
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_chain_cluster_error_merge(_lib.ptr(self.bucket[self.layout.live:]), _lib.current_stream()),
                   "sdumc_chain_cluster_error_merge")

    def adam(self, grad_scale=1.0):
        lib, _lib = self._lib.lib, self._lib
        _lib.check(lib.sdumc_adam_step(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                       _lib.ptr(self.adam_v), self.layout.live, _lib.ptr(self.hyper), self.betas[0],
                                       self.betas[1], self.eps, self.wd, grad_scale, _lib.current_stream()),
                   "sdumc_adam_step")
        _lib.check(lib.sdumc_rng_advance(_lib.ptr(self.rng.t), 2, _lib.current_stream()), "sdumc_rng_advance")
        self.call.next_call()      # (the next forward finds its keep-bits in the set this step's forward filled)

    def set_lr(self, lr):
        self.hyper[0] = lr


class DataParallelStep:
    """`step()` = one optimisation step of the global batch B_global = world_size * B.

    exact=True  : the exchanges of SURVEY §8e -> identical to one process on the whole batch.
    exact=False : local losses (standard DDP semantics: mean of per-shard gradients); NOT
                  comparable to the single-process result because RMSE / RnC are not batch-linear.
    """

    def __init__(self, flat_params, B, T, dims, weights=(0.5, 0.5, 0.1, 0.7, 0.1, 0.8), lr=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=1e-5, seed=0, exact=True, backend_factory=None, bf16=False,
                 force_collectives=False, planes=False):
        import collections
        import inspect
        self.rank, self.world = _world()
        # force_collectives: issue every collective even at world size 1 (a one-rank RCCL communicator): the only way to
        # exercise the RCCL code path -- communicator stream ordering, the async early-slice handle -- on a 1-GPU box.
        self._records = None
        self.collect = self.world > 1 or (bool(force_collectives) and dist.is_initialized())
        self.B, self.exact = B, exact
        self.B_global = B * self.world if exact else B
        factory = backend_factory or HipBackend
        extra = {"bf16": True} if bf16 else {}
        if planes:
            extra["planes"] = True      # (resident batches: HipBackend)
        # One backend per batch shape (B, T_audio, T_text, T_video, T_feat4), least recently used first out, all continuing
        # ONE run state: the reference pads every batch to its own maximum and ends an epoch on a short batch.
        self._shares = "share" in inspect.signature(factory).parameters
        self.state = None
        if self._shares:
            from . import engine
            lay = engine.ParamLayout.get(dims[0], dims[1], dims[2])
            self.state = engine._RunState(flat_params, lay.live, lr, seed)
            extra["share"] = self.state

        def make(Bl, Tl):
            Bg = Bl * self.world if exact else Bl
            return factory(flat_params, Bl, tuple(Tl), dims, weights, lr, betas, eps, weight_decay, seed, self.rank * Bl,
                           Bg, **extra)
        self._make, self._bes, self.max_cached = make, collections.OrderedDict(), 8
        self.be = self._backend(B, T)
        self.weights = weights
        # The early-slice all-reduce is issued asynchronously only on RCCL ("nccl"), where it is a kernel on the
        # communicator's own stream beside the frame-level backward (the pattern torch DDP uses).  Under gloo (CPU tests,
        # the two-ranks-on-one-GPU debugging aid) an in-flight collective stalls every concurrent launch of this process
        # (200 vs 11 ms per step measured), so there the bucket is reduced in one blocking call after the backward.
        # Default: ONE flat all-reduce after a single backward call.  SDUMC_DP_OVERLAP=1 selects the two-slice variant above.
        # On one rank the phase split + the asynchronous hand-off cost 0.075 ms per step (tools/dp_rccl_probe.py), about what
        # the whole 15.4 MB all-reduce should take on an 8 x MI355X xGMI mesh (7 links x ~100 GB/s achievable per GPU), so
        # until it is measured on a multi-GPU node the variant with fewer collectives and no asynchronous hand-off is the default.
        self.overlap = (self.collect and dist.get_backend() == "nccl" and os.environ.get("SDUMC_DP_OVERLAP", "0") == "1")

    def _backend(self, B, T):
        key = (int(B),) + tuple(int(t) for t in T)
        be = self._bes.pop(key, None)
        if be is None:
            if self._bes and not self._shares:
                raise RuntimeError("this compute backend keeps its own optimiser state: one batch shape per run")
            while len(self._bes) >= self.max_cached:
                del self._bes[next(iter(self._bes))]
            be = self._make(B, T)
        self._bes[key] = be
        return be

    def set_batch(self, audio, text, video, feat4, labels, lengths=None):
        """The LOCAL shard: rows [rank*B, (rank+1)*B) of the global batch.  Shapes may change from batch to batch; every
        rank must hold the same local B, and -- for DP(N x B) to equal one process on N*B samples, where the collater pads
        each modality to the GLOBAL batch maximum (read_data.py:223-248) -- the same padded T (`pad_to_global_max`).
        lengths: key-padding extension, forwarded to the kernels (default None = the reference's behaviour)."""
        B, T = audio.shape[0], (audio.shape[1], text.shape[1], video.shape[1], feat4.shape[1])
        self.be = self._backend(B, T)
        self.B, self.B_global = B, (B * self.world if self.exact else B)
        self.be.set_batch(audio, text, video, feat4, labels)
        if lengths is not None or hasattr(self.be, "set_lengths"):
            if hasattr(self.be, "set_lengths"):
                self.be.set_lengths(lengths)
            elif lengths is not None:
                raise RuntimeError("this compute backend has no key-padding extension")

    def pad_to_global_max(self, audio, text, video, feat4):
        """Right-zero-pads the four local feature tensors to the per-modality maximum over all ranks (one tiny MAX
        all-reduce): what the reference's collater does for the whole batch."""
        t = torch.tensor([audio.shape[1], text.shape[1], video.shape[1], feat4.shape[1]], dtype=torch.int64,
                         device=audio.device if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
        if self.collect:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out = []
        for x, T in zip((audio, text, video, feat4), t.tolist()):
            out.append(x if x.shape[1] == T else torch.nn.functional.pad(x, (0, 0, 0, T - x.shape[1])))
        return out

    def load_optimizer_state(self, adam_m, adam_v, step):
        """Resume: Adam moments, step count and dropout call counter of an interrupted run (checkpoint.flat_from_adam_state)."""
        if self.state is None:
            raise RuntimeError("this compute backend keeps its own optimiser state")
        self.state.load_optimizer_state(adam_m, adam_v, step)

    def _gather(self, t):
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous())
        return parts

    def step(self):
        be, B, W = self.be, self.B, self.world
        rnc = be.forward()
        if self.collect and self.exact:
            # ONE collective carries the three exactness exchanges: each rank contributes one record
            # [rnc features (2B x 64) | labels (B) | 3 sums of squares]; RCCL all-gathers the records, while under gloo
            # each rank writes its record into its own row of a zeroed [W, n] buffer that is all-reduced -- an all-gather
            # spelt as a sum with zeros (exact in floating point), because a gloo all_gather of a freshly produced device
            # tensor blocked the host for ~240 ms per call on this stack while all_reduce does not.
            if hasattr(be, "dp_pack"):      # HIP backend: one pack kernel, one collective, one unpack kernel
                rec = be.dp_pack()
                if self._records is None or self._records.shape[1] != rec.numel():
                    self._records = torch.empty(W, rec.numel(), dtype=rec.dtype, device=rec.device)
                if dist.get_backend() == "nccl":
                    dist.all_gather_into_tensor(self._records, rec)
                else:
                    self._records.zero_()
                    self._records[self.rank].copy_(rec)
                    dist.all_reduce(self._records)
                ssd, feats, labels2 = be.dp_unpack(self._records, W)
            else:
                n_f = rnc.numel()
                pack = torch.cat([rnc.reshape(-1), be.labels.reshape(-1).to(rnc.dtype), be.local_ssd().to(rnc.dtype)])
                buf = torch.zeros(W, pack.numel(), dtype=pack.dtype, device=pack.device)
                buf[self.rank] = pack
                dist.all_reduce(buf)
                parts = list(buf)
                fs = [p[:n_f].view(2 * B, -1) for p in parts]   # each (stream 0 rows, stream 1 rows)
                feats = torch.cat([f[:B] for f in fs] + [f[B:] for f in fs]).contiguous()
                lab = torch.cat([p[n_f:n_f + B] for p in parts])
                labels2 = torch.cat([lab, lab]).contiguous()
                ssd = torch.stack([p[n_f + B:n_f + B + 3] for p in parts]).sum(0)   # fixed rank order: same bits on every rank
            losses = be.loss_backward(ssd, feats, labels2, (self.rank * B, W * B + self.rank * B))
        else:
            losses = be.loss_backward()
        # The clustered kernels' error word is per device: a rank whose spin hit its cap must not all-reduce garbage into ranks
        # that then apply it.  Its flag rides in 4 floats behind the gradients (HipBackend.bucket), so the SAME collective
        # tells every rank; err_merge sets the local word wherever the sum is non-zero: no rank applies, every rank raises.
        flagged = self.collect and hasattr(be, "err_flag")
        if self.overlap and hasattr(be, "backward_phase"):
            early = be.backward_phase(0)
            pending = dist.all_reduce(early, async_op=True)   # rides the comm stream beside the frame-level backward
            late = be.backward_phase(1)
            if flagged:
                be.err_flag()
                late = be.bucket[be.layout.early:]
            dist.all_reduce(late)
            pending.wait()
        else:
            grads = be.backward()
            if self.collect:
                if flagged:
                    be.err_flag()
                    grads = be.bucket
                dist.all_reduce(grads)                      # one flat bucket (+ the error flag)
        if flagged:
            be.err_merge()
        be.adam(1.0 if (self.exact or W == 1) else 1.0 / W)
        return losses

    def global_losses(self, losses):
        """[total, mse_full, mse_missing, rmse_text, rmse_query, rmse_fused, rnc] of the GLOBAL batch.  This is the read-back
        point of a data-parallel run, so it is also where a failed clustered-kernel step surfaces: the device's error word
        (include/sdumc_hip.h, sdumc_set_chain_cluster) makes every Adam launch a no-op until it is reset, and is raised here."""
        from . import _lib
        if losses.is_cuda and _lib.lib.sdumc_chain_cluster_error_() > 0:
            raise _lib.SdumcError("a clustered utterance-level kernel ran into its spin cap: the steps since then applied nothing; "
                                  "call sdumc_chain_cluster_reset_error() and repeat them, e.g. with sdumc_set_chain_cluster(0)")
        l = losses.clone()
        if self.collect and self.exact:
            mse = l[1:3].clone()
            dist.all_reduce(mse)                            # MSE terms are local sums / B_global
            l[1:3] = mse
            l[0] = sum(w * v for w, v in zip(self.weights, l[1:7]))
        return l
