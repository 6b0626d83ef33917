"""CPU-only: the C-ABI library loads and exports every symbol include/sdumc_hip.h declares;
host-side helpers that need no GPU (parameter layout, workspace queries) behave."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sdumc_amd import _lib
    header = open(os.path.join(ROOT, "include", "sdumc_hip.h")).read()
    declared = set(re.findall(r"\b(sdumc_[a-z0-9_]+)\s*\(", header))
    assert declared, "header parse failed"
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(_lib.EXPORTS), sorted(declared ^ set(_lib.EXPORTS))
    assert b"gfx950" in _lib.lib.sdumc_version()


def test_struct_sizes_match_the_header():
    """ctypes mirrors vs the C compiler's view of include/sdumc_hip.h."""
    import tempfile
    from sdumc_amd import _lib
    src = r'''
#include <stdio.h>
#include "sdumc_hip.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sdumc_gemm_b1), sizeof(sdumc_gemm_p3), sizeof(sdumc_gemm_bf16),
 sizeof(sdumc_gg_problem), sizeof(sdumc_rows_problem), sizeof(sdumc_umca), sizeof(sdumc_dropout), sizeof(sdumc_gemm),
 sizeof(sdumc_attnpool), sizeof(sdumc_attnpool_bwd_t), sizeof(sdumc_dropsum), sizeof(sdumc_net_dims),
 sizeof(sdumc_net_io), sizeof(sdumc_net_grads), sizeof(sdumc_step_cfg), sizeof(sdumc_softmax), sizeof(sdumc_dropadd),
 sizeof(sdumc_mha), sizeof(sdumc_mha_grads), sizeof(sdumc_gather_seg), sizeof(sdumc_gather_desc)); return 0;}
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(td, "t")]).split()]
    mine = [C.sizeof(c) for c in (_lib.GemmB1, _lib.GemmP3, _lib.GemmBf16, _lib.GGProblem, _lib.RowsProblem, _lib.Umca, _lib.Dropout, _lib.Gemm, _lib.AttnPool, _lib.AttnPoolBwd, _lib.DropSum, _lib.NetDims,
                                  _lib.NetIO, _lib.NetGrads, _lib.StepCfg, _lib.Softmax, _lib.DropAdd, _lib.Mha,
                                  _lib.MhaGrads, _lib.GatherSeg, _lib.GatherBatch)]
    assert sizes == mine


def test_param_table_and_workspace_queries():
    from sdumc_amd import _lib
    from sdumc_amd.engine import ParamLayout, make_dims
    from oracle import sdumc_oracle as O
    lay = ParamLayout.get(1024, 4096, 1024)
    shapes = O.param_shapes((1024, 4096, 1024, 4096))
    assert {k: tuple(v[1]) for k, v in lay.entries.items()} == {k: tuple(v) for k, v in shapes.items()}
    assert lay.total == 4268884 + 16 and lay.live == 3857291 + 5
    offs = sorted((v[0], int(np.prod(v[1]))) for v in lay.entries.values())
    for (o0, n0), (o1, _) in zip(offs, offs[1:]):
        assert o0 + n0 <= o1 and o1 % 4 == 0
    d = make_dims(64, 2, 375, 225, (32, 32), (1024, 4096, 1024), True)
    n = _lib.lib.sdumc_net_workspace_bytes(C.byref(d))
    s = _lib.lib.sdumc_step_workspace_bytes(C.byref(d))
    assert 0 < n < s < 4 << 30
    bad = make_dims(0, 2, 375, 225, (32, 32), (1024, 4096, 1024), True)
    assert _lib.lib.sdumc_net_workspace_bytes(C.byref(bad)) == 0
    # bad arguments come back as error codes, never as exceptions across the ABI
    assert _lib.lib.sdumc_gemm_f32(None, None) == -1
    assert _lib.lib.sdumc_net_forward(C.byref(d), None, None) == -1


def test_product_path_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "sdumc_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(root, f)).read(), f


def test_collate_matches_reference_golden(golden):
    """sdumc_amd.data.collate reproduces pad_to_maxlen_pre_modality_tensor_4 + the batch tuple."""
    import torch
    from sdumc_amd.data import collate
    g = golden("collate")
    n = len(g["lens"])
    inst = [{k: torch.from_numpy(g[f"raw_{k}_{b}"]) for k in ("audio", "text", "video", "feat4")} | {"emo": 0, "val": 0.5 * b, "name": f"u{b}"}
            for b in range(n)]
    batch, pads, emos, vals, names = collate(inst)
    for k in ("audios", "texts", "videos", "feat4s"):
        np.testing.assert_array_equal(batch[k].numpy(), g[k])
    np.testing.assert_array_equal(np.array(pads), g["pads"])
    assert names == ["u0", "u1", "u2"] and vals.tolist() == [0.0, 0.5, 1.0] and emos.shape == (n,)


def test_schedule_and_metric(golden):
    from sdumc_amd.schedule import warm_up_with_step_lr
    from sdumc_amd.metric import eval_mosei_metric
    np.testing.assert_allclose([warm_up_with_step_lr(e) for e in range(40)], golden("step")["lr_table"], rtol=1e-12)
    from sklearn.metrics import f1_score, accuracy_score, mean_absolute_error
    rs = np.random.RandomState(0)
    y = np.round(rs.uniform(-3, 3, 200), 1); y[:10] = 0
    p = y + rs.normal(0, 1.0, 200)
    m = eval_mosei_metric(p, y)
    nz = y != 0
    np.testing.assert_allclose(m["mae"], mean_absolute_error(y, p))
    np.testing.assert_allclose(m["acc2"], accuracy_score(y[nz] > 0, p[nz] > 0))
    np.testing.assert_allclose(m["f1"], f1_score(y[nz] > 0, p[nz] > 0, average="weighted"))


def test_run_state_resume_installs_step_count_and_dropout_counter():
    """engine._RunState.load_optimizer_state (host logic, no kernel call): moments copied, hyper[1] = step (drives the Adam
    bias correction in the fused step), Philox call counter = 2 * step (two forward calls per optimisation step)."""
    import torch
    from sdumc_amd import engine
    st = engine._RunState(torch.zeros(16), 8, 1e-3, seed=(7 << 32) | 5)
    m, v = torch.arange(8.0), torch.arange(8.0) * 2
    st.load_optimizer_state(m, v, 11)
    assert torch.equal(st.adam_m, m) and torch.equal(st.adam_v, v)
    assert float(st.hyper[0]) == pytest.approx(1e-3) and float(st.hyper[1]) == 11.0
    assert st.rng.call == 22 and st.rng.t[:2].tolist() == [5, 7]
    assert st.optimizer_state()[2] == 11
    with pytest.raises(Exception):
        st.load_optimizer_state(torch.zeros(4), v, 1)
