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


def unpack(data, device):
    """What train_or_eval_model reads from one batch tuple (main :94-109), moved to `device`."""
    b = data[0]
    return (b['audios'].to(device, non_blocking=True), b['texts'].to(device, non_blocking=True),
            b['videos'].to(device, non_blocking=True), b['feat4s'].to(device, non_blocking=True),
            data[-2].float().to(device, non_blocking=True), data[-1])
