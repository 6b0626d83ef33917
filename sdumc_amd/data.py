"""The feature-tensor (dataloader) contract of the hot path:

  data = (batch_dict, pads, emos, vals, names)       toolkit/data/feat_data.py:232-253
  batch_dict['audios'|'texts'|'videos'|'feat4s']     float32 [B, maxT_m, d_m], RIGHT-zero-padded per
                                                     modality to the batch max (read_data.py:139-151,223-248)
  pads   4 lists of per-sample pad lengths (unused downstream), emos/vals float [B], names list[str]
  on disk: <feat_root>/<feature_name>/<utt>.npy of shape [T, d] (squeezed; 1-D promoted to [1, d])
           (read_data.py:22-49)

Padded frames are NOT masked anywhere on the reference's path: they take part in the softmax over time.
This module reproduces the contract (not the reference's 12-process loader)."""
import os

import numpy as np
import torch

KEYS = ('audios', 'texts', 'videos', 'feat4s')


def read_feature(feature_root, name):
    """[T, d] float32 from <root>/<name>.npy or a directory of per-frame .npy files (read_data.py:22-49)."""
    path = os.path.join(feature_root, name + '.npy')
    if os.path.exists(path):
        feat = np.load(path).squeeze()
    elif os.path.isdir(os.path.join(feature_root, name)):
        d = os.path.join(feature_root, name)
        feat = np.array([np.load(os.path.join(d, f)) for f in sorted(os.listdir(d))]).squeeze()
    else:
        raise FileNotFoundError(path)
    if feat.ndim == 1:
        feat = feat[np.newaxis, :]
    return np.ascontiguousarray(feat, dtype=np.float32)


def pad_right(feats):
    """list of [T_i, d] tensors -> ([B, maxT, d], pad_lens)  (pad_to_maxlen_pre_modality_tensor_4)"""
    lens = [int(f.shape[0]) for f in feats]
    mx = max(lens)
    out = torch.zeros(len(feats), mx, feats[0].shape[1], dtype=torch.float32)
    for i, f in enumerate(feats):
        out[i, :lens[i]] = torch.as_tensor(f, dtype=torch.float32)
    return out, [mx - n for n in lens]


def collate(instances):
    """instances: dicts with 'audio','text','video','feat4' ([T,d]), 'emo','val','name'
    (Data_Feat_MOSEI_EmoVal_4F.__getitem__, feat_data.py:218-229) -> the reference's batch tuple."""
    batch, pads = {}, []
    for key, src in zip(KEYS, ('audio', 'text', 'video', 'feat4')):
        batch[key], p = pad_right([inst[src] for inst in instances])
        pads.append(p)
    emos = torch.FloatTensor([inst['emo'] for inst in instances])
    vals = torch.FloatTensor([inst['val'] for inst in instances])
    names = [inst['name'] for inst in instances]
    return batch, pads, emos, vals, names


def lengths_from_pads(data):
    """Valid frame counts (audio, text, video, feat4) of one batch tuple = maxT_m - pad_len (the `pads` the reference's
    collater returns and never uses, feat_data.py:244-253): what `lengths=` of the key-padding extension takes."""
    b, pads = data[0], data[1]
    return tuple(torch.tensor([b[k].shape[1] - int(p) for p in pad], dtype=torch.int32) for k, pad in zip(KEYS, pads))


def unpack(data, device):
    """What train_or_eval_model reads from one batch tuple (main :94-109), moved to `device`."""
    b = data[0]
    return (b['audios'].to(device, non_blocking=True), b['texts'].to(device, non_blocking=True),
            b['videos'].to(device, non_blocking=True), b['feat4s'].to(device, non_blocking=True),
            data[-2].float().to(device, non_blocking=True), data[-1])


class EpochPlan:
    """The batches of one epoch as the device sees them: ONE index tensor uploaded once (no per-batch host -> device copy inside the
    loop -- a pageable upload per step costs the host a synchronisation with the previous step), the host-side offsets into it and the
    padded shape (B, (T_audio, T_text, T_video, T_feat4)) of every batch."""

    def __init__(self, idx_d, offsets, shapes):
        self.idx_d, self.offsets, self.shapes = idx_d, offsets, shapes

    def __len__(self):
        return len(self.shapes)

    def idx_ptr(self, i):
        return self.idx_d.data_ptr() + 8 * self.offsets[i]


class DeviceFeatureStore:
    """All pre-extracted features of a split, packed per modality into ONE device tensor [sum T, d]
    (MOSEI train at WavLM/Vicuna/MANet widths is tens of GB: it fits the 288 GB of one MI355X many times).
    `batch(indices)` assembles the reference's batch tuple on the GPU with a HIP gather/pad kernel
    (sdumc_gather_pad), so no feature bytes cross PCIe inside the training loop.  Replaces the roles of
    Data_Feat_MOSEI_EmoVal_4F.collater + pad_to_maxlen_pre_modality_tensor_4 (feat_data.py:232-253,
    read_data.py:223-248) for the hot path; the on-disk format is the reference's (read_feature).

    Every packed tensor ends in ONE all-zero row: a step can then read a batch IN PLACE through a row map (sdumc_net_io.row_map; entry
    of a padded frame = that row) instead of from a padded copy -- gather_desc(maps_out=...) writes the maps, 4 bytes per frame.

    planes=True (fp32 storage): every utterance is ALSO held as P3 planes (sdumc_p3_split of the packed tensor, three bf16 parts per
    value, 6 d bytes per row) -- split ONCE per dataset, where the reference re-reads the feature files every epoch; the store then
    takes 2.5x the fp32 bytes (the fp32 rows the weight gradients read + 1.5x for the planes the frame projections read).  A batch's
    planes are gathered like its fp32 rows (a padded row is zero in both, so this equals splitting the gathered batch bit for bit)."""

    MODS = ('audio', 'text', 'video', 'feat4')

    def __init__(self, instances, device='cuda', bf16=False, planes=False):
        """instances: iterable of dicts with 'audio','text','video','feat4' ([T, d] arrays), 'emo', 'val', 'name'.
        bf16=True holds the features as bf16 (half the HBM; what the engine's bf16-storage mode reads; widths % 8 == 0)."""
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        instances = list(instances)
        self.device = torch.device(device)
        self.names = [inst['name'] for inst in instances]
        self.vals = torch.tensor([float(inst['val']) for inst in instances], dtype=torch.float32, device=self.device)
        self.emos = torch.tensor([float(inst['emo']) for inst in instances], dtype=torch.float32, device=self.device)
        self.packed, self.start, self.length, self.dim = {}, {}, {}, {}
        for m in self.MODS:
            lens = [int(inst[m].shape[0]) for inst in instances]
            d = int(instances[0][m].shape[1])
            if d % 4:
                raise _lib.SdumcError(f"feature width {d} of '{m}' must be a multiple of 4")
            starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            host = torch.zeros(sum(lens) + 1, d, dtype=torch.float32)      # (+ one all-zero row: what a padded frame's row-map entry names)
            for s, n, inst in zip(starts, lens, instances):
                host[s:s + n] = torch.as_tensor(inst[m], dtype=torch.float32)
            self.packed[m] = host.to(self.device).to(torch.bfloat16 if bf16 else torch.float32)
            if bf16 and d % 8:
                raise _lib.SdumcError(f"bf16 feature width {d} of '{m}' must be a multiple of 8")
            self.start[m] = torch.from_numpy(starts)
            self.length[m] = torch.tensor(lens, dtype=torch.int32)
            self.dim[m] = d
        self._device_tables()
        self.packed_p3 = None
        if planes:
            self.make_planes()

    def _device_tables(self):
        # store-wide tables on the device: a batch is then named by its index vector alone (sdumc_gather_batch)
        self.start_d = {m: self.start[m].to(self.device) for m in self.MODS}
        self.length_d = {m: self.length[m].to(self.device) for m in self.MODS}
        self._len_np = {m: self.length[m].numpy() for m in self.MODS}

    def make_planes(self):
        """The P3 planes of every packed tensor (fp32 storage, widths in whole 64-element k-tiles: what csrc/gemm_p3.hip reads)."""
        _lib = self._lib
        if any(t.dtype != torch.float32 for t in self.packed.values()):
            raise _lib.SdumcError("planes: the store holds bf16 features (the bf16-storage step reads them as they are)")
        if any(self.dim[m] % 64 for m in self.MODS):
            raise _lib.SdumcError("planes: feature widths must be multiples of 64")
        self.packed_p3 = {}
        for m in self.MODS:
            src, d = self.packed[m], self.dim[m]
            dst = torch.empty(src.shape[0], 6 * d, dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.sdumc_p3_split(_lib.ptr(src), d, _lib.ptr(dst), 6 * d, src.shape[0], d, _lib.current_stream()), "sdumc_p3_split")
            self.packed_p3[m] = dst
        return self

    @property
    def nbytes(self):
        n = sum(t.numel() * t.element_size() for t in self.packed.values())
        if self.packed_p3 is not None:
            n += sum(t.numel() for t in self.packed_p3.values())
        return n

    @classmethod
    def synthetic(cls, n, T, dims, seed=1234, device='cuda', min_frac=0.25, bf16=False, planes=False):
        """n utterances with per-sample lengths ~ U{ceil(min_frac * T_m) .. T_m} and N(0, 1) features, generated on the device
        (SURVEY §8d's variable-length synthetic inputs; no host copy of the tens of GB a real split holds)."""
        import ctypes as C
        from . import _lib
        self = cls.__new__(cls)
        self._C, self._lib = C, _lib
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        gd = torch.Generator(device=self.device).manual_seed(seed)
        self.names = [f"utt{i:06d}" for i in range(n)]
        self.vals = (torch.rand(n, generator=g) * 6 - 3).to(self.device)
        self.emos = torch.zeros(n, device=self.device)
        self.packed, self.start, self.length, self.dim = {}, {}, {}, {}
        for m, Tm, d in zip(self.MODS, T, dims):
            lo = max(1, int(np.ceil(Tm * min_frac)))
            lens = torch.randint(lo, Tm + 1, (n,), generator=g, dtype=torch.int32)
            starts = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens.to(torch.int64), 0)[:-1]])
            rows = int(lens.sum())
            self.packed[m] = torch.zeros(rows + 1, d, device=self.device, dtype=torch.bfloat16 if bf16 else torch.float32)      # (+ the zero row)
            self.packed[m][:rows] = torch.randn(rows, d, device=self.device, generator=gd)
            self.start[m], self.length[m], self.dim[m] = starts, lens, int(d)
        self._device_tables()
        self.packed_p3 = None
        if planes:
            self.make_planes()
        return self

    def batch_shape(self, indices):
        """(B, (T_audio, T_text, T_video, T_feat4)) of the padded batch -- host-side table lookups only."""
        idx = self._checked(indices)
        return int(idx.numel()), tuple(int(self.length[m][idx].max()) for m in self.MODS)

    def _checked(self, indices):
        """host-side range check: the gather kernel indexes device tables with these values"""
        idx = torch.as_tensor(indices, dtype=torch.int64).reshape(-1)
        n = len(self)
        if idx.numel() == 0 or int(idx.min()) < 0 or int(idx.max()) >= n:
            raise self._lib.SdumcError(f"sample index out of range [0, {n})")
        return idx

    def plan_epoch(self, batches):
        """batches: the epoch's index vectors (what a BatchSampler yields) -> EpochPlan: every index range-checked, one upload."""
        idxs = [self._checked(b) for b in batches]
        if not idxs:
            raise self._lib.SdumcError("plan_epoch: no batches")
        offsets = np.concatenate([[0], np.cumsum([i.numel() for i in idxs])]).astype(np.int64)
        shapes = []
        for i in idxs:
            ii = i.numpy()
            shapes.append((int(ii.size), tuple(int(self._len_np[m][ii].max()) for m in self.MODS)))
        return EpochPlan(torch.cat(idxs).to(self.device), [int(o) for o in offsets[:-1]], shapes)

    def gather_desc(self, idx_ptr, B, T, outs, labels_out, lengths_out=None, planes_out=None, maps_out=None):
        """The sdumc_gather_batch descriptor of one batch: idx_ptr = device address of its int64 [B] index vector, T its padded frame
        counts, outs = 4 device buffers of >= B * T_m * d_m elements (fp32, or bf16 for a bf16 store), planes_out = 4 uint8 buffers of
        >= B * T_m * 6 d_m bytes (needs the store's planes), labels_out [>= B], lengths_out = optional 4 int32 [>= B].
        maps_out = 4 int32 buffers of >= B * T_m entries INSTEAD of outs / planes_out (pass outs=None): no padded copy is made, the
        step reads the store in place through these row maps (sdumc_net_io.row_map)."""
        _lib = self._lib
        if planes_out is not None and self.packed_p3 is None:
            raise _lib.SdumcError("this store holds no planes (DeviceFeatureStore(planes=True))")
        if (outs is None) == (maps_out is None):
            raise _lib.SdumcError("gather_desc: padded copies (outs) or row maps (maps_out), one of the two")
        g = _lib.GatherBatch()
        n = 0
        for k, m in enumerate(self.MODS):
            if maps_out is not None:
                sg = g.seg[n]
                sg.start_all, sg.len_all = _lib.ptr(self.start_d[m]), _lib.ptr(self.length_d[m])
                sg.map_out, sg.zero_row = _lib.ptr(maps_out[k]), int(self.packed[m].shape[0]) - 1
                sg.len_out = _lib.ptr(lengths_out[k]) if lengths_out is not None else None
                sg.Tmax, sg.d4 = int(T[k]), 0
                n += 1
                continue
            srcs = [(self.packed[m], outs[k], self.dim[m] * self.packed[m].element_size() // 16)]
            if planes_out is not None:
                srcs.append((self.packed_p3[m], planes_out[k], 6 * self.dim[m] // 16))
            for j, (src, dst, d4) in enumerate(srcs):
                sg = g.seg[n]
                sg.packed, sg.start_all, sg.len_all = _lib.ptr(src), _lib.ptr(self.start_d[m]), _lib.ptr(self.length_d[m])
                sg.out = _lib.ptr(dst)
                sg.len_out = _lib.ptr(lengths_out[k]) if (lengths_out is not None and j == 0) else None
                sg.Tmax, sg.d4 = int(T[k]), int(d4)
                n += 1
        g.nseg, g.B, g.idx = n, int(B), idx_ptr
        g.labels_all, g.labels_out = _lib.ptr(self.vals), _lib.ptr(labels_out)
        return g

    def batch_into(self, indices, outs, labels_out, lengths_out=None, planes_out=None):
        """Assembles the batch `indices` into caller-owned buffers: outs = 4 device tensors [B, Tmax_m, d_m] (e.g. the input
        buffers of an engine.TrainStep), labels_out [B]; lengths_out = optional 4 int32 device tensors (>= B) that receive the
        valid frame counts; planes_out = optional 4 uint8 tensors (>= B * Tmax_m * 6 d_m bytes) that receive the batch's P3 planes.
        One index-vector upload, ONE gather launch, nothing else."""
        _lib = self._lib
        idx = self._checked(indices)
        B = idx.numel()
        idx_d = idx.to(self.device, non_blocking=True)
        for k, m in enumerate(self.MODS):
            dst = outs[k]
            if dst.shape[0] != B or dst.shape[2] != self.dim[m] or not dst.is_contiguous() or dst.dtype != self.packed[m].dtype:
                raise _lib.SdumcError("batch_into: output buffer does not match the batch (shape / dtype)")
            if planes_out is not None and (planes_out[k].dtype != torch.uint8 or planes_out[k].numel() < B * dst.shape[1] * 6 * self.dim[m]):
                raise _lib.SdumcError("batch_into: planes buffer too small")
        g = self.gather_desc(idx_d.data_ptr(), B, [o.shape[1] for o in outs], outs, labels_out, lengths_out, planes_out)
        _lib.check(_lib.lib.sdumc_gather_batch(self._C.byref(g), 0, _lib.current_stream()), "sdumc_gather_batch")
        self._keep_idx = idx_d
        return lengths_out

    def __len__(self):
        return len(self.names)

    def get_featdim(self):
        return tuple(self.dim[m] for m in self.MODS)          # (adim, tdim, vdim, f4dim), feat_data.py:256-258

    def batch(self, indices):
        """-> (batch_dict, pads, emos, vals, names) with device tensors, identical to collate() of the same samples."""
        C, _lib = self._C, self._lib
        idx = torch.as_tensor(indices, dtype=torch.int64)
        B = idx.numel()
        out, pads = {}, []
        for key, m in zip(KEYS, self.MODS):
            lens = self.length[m][idx]
            tmax = int(lens.max())
            start_d = self.start[m][idx].to(self.device)
            len_d = lens.to(self.device)
            dst = torch.empty(B, tmax, self.dim[m], dtype=self.packed[m].dtype, device=self.device)
            dw = self.dim[m] if dst.dtype == torch.float32 else self.dim[m] // 2
            _lib.check(_lib.lib.sdumc_gather_pad(_lib.ptr(self.packed[m]), _lib.ptr(start_d), _lib.ptr(len_d), B, tmax,
                                                 dw, _lib.ptr(dst), _lib.current_stream()), "sdumc_gather_pad")
            out[key] = dst
            pads.append((tmax - lens).tolist())
        idx_d = idx.to(self.device)
        return out, pads, self.emos[idx_d], self.vals[idx_d], [self.names[i] for i in idx.tolist()]
