"""CPU: host-side logic that needs no GPU — C-ABI export check, config plumbing, user sharding,
triplet sampler, split counts, loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "yelprec_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(yr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/yelprec_engine.h declares
    (no compute calls here: there is no GPU in this container)."""
    from yelprecommendation_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.yr_engine_version() == _lib.ENGINE_VERSION
    assert lib.yr_engine_arch() == b"gfx950"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    assert raw.yr_engine_version() == _lib.ENGINE_VERSION


def test_header_version_matches_binding():
    from yelprecommendation_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "yelprec_engine.h")).read()
    assert int(re.search(r"#define YR_ENGINE_VERSION (\d+)", hdr).group(1)) == _lib.ENGINE_VERSION
    assert int(re.search(r"#define YR_LOSS_PARTIALS (\d+)", hdr).group(1)) == 2048


def test_argument_checks_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from yelprecommendation_amd import _lib
    lib = _lib.load()
    assert lib.yr_bpr_mf_pull_workspace_bytes(-1, 10, 10, 64) == -2
    assert lib.yr_bpr_mf_pull_workspace_bytes(10, 10, 10, 48) == -1
    n = lib.yr_bpr_mf_pull_workspace_bytes(1 << 20, 31668, 38048, 64)
    assert 40e6 < n < 49e6                       # 36 B per triplet + per-tile bucket offsets + 4 MB of split-bucket scratch
    # a size computed for max_batch serves every smaller batch (the tiling depends on B)
    for b in (0, 1, 1000, 1 << 15, (1 << 15) + 1, 65536, 200000, (1 << 20) - 1):
        assert lib.yr_bpr_mf_pull_workspace_bytes(b, 31668, 38048, 64) <= n
    assert lib.yr_adam_dense(None, None, None, None, 16, 1e-3, 1e-3, 1.0, 0.9, 0.999, 1e-8, 0.0, 0, 0, None) == -2
    assert lib.yr_adam_dense(None, None, None, None, 0, 1e-3, 1e-3, 1.0, 0.9, 0.999, 1e-8, 0.0, 0, 0, None) == 0
    assert lib.yr_topk_masked(None, 1, 10, 10, None, None, None, 0.0, 100, None, None) == -2   # k > 64
    # yr_adam_dense_dual with row marks: the lanes of a row must share a wave (row_width / 4 divides 64), as
    # yr_adam_dense_flat requires — 48- and 96-wide rows are refused before any launch (fake aligned addresses)
    a = 4096
    for rw in (48, 96, 512):
        assert lib.yr_adam_dense_dual(a, a, a, a, 4 * rw, a, a, a, a, 4 * rw, rw, a, a, 1e-3, 1e-3, 1.0, 0.9, 0.999,
                                      1e-8, 0.0, 0, None, 1.0, None, None, None) == -2
    # entry points added in round 2: sizes / helpers answer on the host, bad arguments come back as -2 / -1
    assert lib.yr_bpr_mf_pull_item_buckets(38048, 64) == 2378 and lib.yr_bpr_mf_pull_item_buckets(38048, 48) == -2
    assert lib.yr_cdae_sparse_part_columns(38048) == 1192 and lib.yr_cdae_decode_loss_partials(256, 38048) == 4 * 595
    assert [lib.yr_cdae_sampled_decode_splits(b) for b in (0, 32, 256, 1024)] == [1, 8, 2, 1]
    assert lib.yr_gemm_f32_ex(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None, None, None) == -2
    assert lib.yr_gemm_f32_ex(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None, None, None) == 0    # empty
    assert lib.yr_cdae_decode_loss(None, None, None, None, None, 8, 8, 6, 1, None, 8, None, None, None, None) == -2
    assert lib.yr_cdae_sampled_decode(None, None, None, None, None, None, 8, 8, 300, 1, None, None, None, None, None,
                                      None) == -1                                                   # H > 256
    assert lib.yr_cdae_train_lists(None, None, None, None, None, 4, 4, 1 << 20, 5, 1, 2, 0.5, None, None, None, None,
                                   None, None, None, None) == -2
    assert lib.yr_cdae_train_lists(None, None, None, None, None, 0, 4, 10, 5, 1, 2, 0.5, None, None, None, None, None,
                                   None, None, None) == 0                                           # empty batch
    assert lib.yr_adam_dense_flat(None, None, None, None, None, None, None, None, None, None, 17, 1e-3, 1e-3, 1.0, 0.9,
                                  0.999, 1e-8, 0.0, 0, None) == -2                                  # > 16 tensors
    assert lib.yr_adam_dense_flat(None, None, None, None, None, None, None, None, None, None, 0, 1e-3, 1e-3, 1.0, 0.9,
                                  0.999, 1e-8, 0.0, 0, None) == 0
    assert lib.yr_mf_eval_topk_bias(None, None, None, None, 4, 64, 10, 10, None, None, 0.0, 33, None, None, 0, 0, None,
                                    None, None) == -2                                               # k > 32
    assert lib.yr_mf_eval_topk_bias(None, None, None, None, 4, 128, 10, 10, None, None, 0.0, 20, None, None, 0, 0, None,
                                    None, None) == -1                                               # k > 16 at D = 128: unsupported
    assert lib.yr_mf_eval_topk_bias(None, None, None, None, 4, 64, 10, 10, None, None, 0.0, 10, None, None, 0, 8, None,
                                    None, None) == -2                                               # unknown mode bit
    assert lib.yr_mf_eval_topk_bias(None, None, None, None, 4, 64, 10, 10, None, None, 0.0, 10, None, None, 0, 6, None,
                                    None, None) == -2                                               # prescan off AND forced
    assert lib.yr_mf_eval_topk_planes_bytes(1000, 64) == 1000 * 6 * 64
    assert lib.yr_mf_eval_topk_planes_bytes(1000, 48) == -2
    assert lib.yr_mf_eval_topk_workspace_bytes(128, 1000, 64, 10, 1) == 1000 * 6 * 64 + 512         # one slice: planes + row thresholds
    assert lib.yr_mf_eval_topk_workspace_bytes(128, 1000, 64, 10, 0) == 512
    assert lib.yr_mf_eval_topk_workspace_bytes(128, 1000, 64, 10, 4) == 512 + 128 * 128             # forced prescan: maxima
    assert lib.yr_mf_eval_topk_workspace_bytes(128, 20000, 64, 10, 0) > lib.yr_mf_eval_topk_workspace_bytes(128, 20000, 64, 10, 2)
    assert lib.yr_mf_eval_topk_workspace_bytes(128, 20000, 64, 4, 0) == lib.yr_mf_eval_topk_workspace_bytes(128, 20000, 64, 4, 2)


def test_ops_refuse_cpu_tensors():
    from yelprecommendation_amd import engine
    from yelprecommendation_amd._lib import EngineError
    z = torch.zeros(8, 64)
    i = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(EngineError, match="no CPU fallback"):
        engine.mf_score(z, z, i, i)
    with pytest.raises(EngineError):
        engine.adam_dense(z, z, z, z, 1, 1e-3)


def test_model_on_cpu_fails_loudly(tmp_path):
    """cfg.device='cpu' is still a legal name (reference default) but nothing computes there."""
    from yelprecommendation_amd._lib import EngineError
    from yelprecommendation_amd.trainers import MFTrainer
    from yelprecommendation_amd.utils import make_config
    t = MFTrainer(make_config("MF", device="cpu", model_dir=str(tmp_path)), 20, 30)
    assert sorted(t.model.state_dict()) == ["item_embedding.weight", "user_embedding.weight"]
    with pytest.raises(EngineError):
        t.model(torch.zeros(2, dtype=torch.int64), torch.zeros(2, dtype=torch.int64))


def test_config_unpack_model():
    from yelprecommendation_amd.utils import DEFAULTS, Config, make_config, unpack_model
    cfg = make_config("NGCF", lr=0.01)
    assert cfg.embed_size == 64 and cfg.num_orders == 2 and cfg.lr == 0.01 and "model" not in cfg
    assert make_config("CDAE").hidden_size == 64 and make_config("CDAE").corruption_level == 0.6
    bad = Config(DEFAULTS)
    bad["model_name"] = "nope"
    with pytest.raises(ValueError):
        unpack_model(bad)


def test_base_trainer_surface(tmp_path):
    from yelprecommendation_amd.trainers import MFTrainer
    from yelprecommendation_amd.utils import make_config
    t = MFTrainer(make_config("MF", device="cpu", model_dir=str(tmp_path / "m"), best_metric="recall"), 5, 7)
    assert os.path.isdir(str(tmp_path / "m"))
    assert t._is_surpass_best_metric(current=(1.0, 0, 0.5, 0, 0), best=(0.5, 0, 0.4, 0, 0))
    assert not t._is_surpass_best_metric(current=(0.1, 0, 0.3, 0, 0), best=(0.5, 0, 0.4, 0, 0))
    with pytest.raises(NotImplementedError, match="Not implemented model: a"):
        t._model("a")                                   # reference test/trainers/test_base_trainer.py intent
    with pytest.raises(NotImplementedError):
        t._optimizer("rmsprop", t.model, 1e-3)
    assert t._device("tpu") == torch.device("cpu")
    for name in ("adam", "adamw", "sgd"):
        assert t._optimizer(name, t.model, 1e-3, 0.0) is not None


def test_user_shard_partition():
    from yelprecommendation_amd.user_shard import UserShard
    for n, w in ((31668, 8), (10, 3), (7, 8), (64, 1)):
        shards = [UserShard(n, w, r) for r in range(w)]
        assert shards[0].lo == 0 and shards[-1].hi == n
        assert all(a.hi == b.lo for a, b in zip(shards, shards[1:]))
        assert max(s.size for s in shards) - min(s.size for s in shards) <= 1
        ids = np.arange(n)
        owner = shards[0].owner(ids)
        for s in shards:
            assert (owner[s.lo:s.hi] == s.rank).all()
        t_owner = shards[0].owner(torch.arange(n))
        assert (t_owner.numpy() == owner).all()
    s = UserShard(10, 3, 1)
    u = np.array([0, 4, 5, 6, 9]); p = np.array([1, 2, 3, 4, 5])
    lu, lp = s.select(u, p)
    assert lu.tolist() == [0, 1, 2] and lp.tolist() == [2, 3, 4]


def test_split_counts_and_sampler_refuses_cpu():
    from yelprecommendation_amd._lib import EngineError
    from yelprecommendation_amd.data.synthetic import make_interactions_torch
    from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
    u, i = make_interactions_torch(400, 300, 12.0, seed=3)
    assert int(u.max()) == 399 and int(i.max()) < 300 and (torch.bincount(u) >= 5).all()
    key = u * 300 + i
    assert key.unique().numel() == key.numel() and (key[1:] > key[:-1]).all()
    label = split_train_rows(u, i)
    cnt = torch.bincount(u)
    for user in (0, 17, 399):
        n = int(cnt[user]); lab = label[u == user]
        n_test = int(np.ceil(0.2 * n)); n_valid = int(np.ceil(0.25 * (n - n_test)))
        assert int((lab == 2).sum()) == n_test and int((lab == 1).sum()) == n_valid   # sklearn's counts
    tr = label == 0
    with pytest.raises(EngineError):                 # the sampler is a HIP kernel: no CPU fallback
        TripletSampler(u[tr], i[tr], 400, 300, seed=1)


def test_numpy_generator_shape():
    from yelprecommendation_amd.data.synthetic import make_interactions
    u, i, r = make_interactions(300, 250, 10.0, seed=5, min_item_degree=5)
    assert u.min() == 0 and u.max() == 299 and i.max() == 249 and set(np.unique(r)) <= {1, 2, 3, 4, 5}
    assert np.bincount(i).min() >= 5 and np.bincount(u).min() >= 5
    assert len(set(zip(u.tolist(), i.tolist()))) == len(u)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: with the shared library absent, loading raises and names the build command."""
    from yelprecommendation_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(str(tmp_path), "libyelprec_engine.so"))
    monkeypatch.setattr(_lib, "_lib", None)                          # forget the cached handle for this test
    with pytest.raises(_lib.EngineError, match="not found"):
        _lib.load()


def test_bf16x3_split_reproduces_f32_products():
    """The arithmetic behind YR_EVAL_BF16X3 (csrc/eval_topk.hip, et_split3), restated in NumPy: three bfloat16 terms
    (round to nearest even each time, residuals exact in f32) reproduce an f32 value to 2^-24 of its magnitude or
    better, and the six partial products of weight >= 2^-18 give every dot product to well below the f32 rounding
    of the same dot product (compared with the exact float64 value)."""
    def bf16(x):                                     # f32 -> nearest-even bfloat16, returned as f32
        b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        b = ((b + 0x7fff + ((b >> 16) & 1)) >> 16) << 16
        return b.astype(np.uint32).view(np.float32)

    def split3(x):
        x1 = bf16(x)
        r1 = (x - x1).astype(np.float32)
        x2 = bf16(r1)
        x3 = bf16((r1 - x2).astype(np.float32))
        return x1, x2, x3

    rs = np.random.RandomState(3)
    x = (rs.standard_normal(20000) * np.exp(rs.uniform(-20, 20, 20000))).astype(np.float32)
    x1, x2, x3 = split3(x)
    assert np.all(np.abs(x.astype(np.float64) - x1 - x2.astype(np.float64) - x3) <= np.abs(x) * 2.0 ** -24)
    U = (rs.standard_normal((64, 64)) * 0.1).astype(np.float32)
    I = (rs.standard_normal((512, 64)) * 0.1).astype(np.float32)
    u, i = [p.astype(np.float64) for p in split3(U)], [p.astype(np.float64) for p in split3(I)]
    six = sum(u[a] @ i[b].T for a, b in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))
    exact = U.astype(np.float64) @ I.astype(np.float64).T
    scale = np.abs(U).astype(np.float64) @ np.abs(I).astype(np.float64).T        # sum_d |u_d i_d|
    assert np.max(np.abs(six - exact) / scale) < 2.0 ** -24                      # dropped terms: x2 y3, x3 y2, x3 y3
    f32 = (U @ I.T).astype(np.float64)                                           # what an f32 dot product rounds to
    assert np.max(np.abs(six - exact) / scale) < np.max(np.abs(f32 - exact) / scale)
    five = six - u[1] @ i[1].T                                                   # one product fewer is NOT enough
    assert np.max(np.abs(five - exact) / scale) > 2.0 ** -22


def test_kernel_resource_budgets_hold_and_the_guard_bites():
    """scripts/kernel_resources.py (called by __graft_entry__.build()): the register / LDS / scratch figures of the
    built gfx950 code objects stay inside the occupancy budgets the design depends on (evaluation sweep <= 168 VGPRs:
    three waves per SIMD; owner passes <= 64 VGPRs and <= 20 KB LDS: every bucket's workgroup resident at once), and
    the check does fail when a kernel goes over."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kernel_resources as kr
    ks = kr.check()
    sweep = [r for n, r in ks.items() if n.startswith("void yr::mf_eval_topk_kernel<64, 10, false, true, false>")]
    owner = [r for n, r in ks.items() if n.startswith("void yr::owner_pass_kernel<64, true, true, false, 0>")]
    assert len(sweep) == 1 and len(owner) == 1 and sweep[0]["vgpr"] <= 168 and owner[0]["vgpr"] <= 64
    # no scratch TRAFFIC (a private segment may be reserved without one instruction touching it)
    assert sweep[0]["scratch_ops"] == 0 and owner[0]["scratch_ops"] == 0 and owner[0]["spill"] == 0
    fat = {n: dict(r, vgpr=r["vgpr"] + (24 if "mf_eval_topk_kernel<64, 10, false, true, false>" in n else 0)) for n, r in ks.items()}
    with pytest.raises(RuntimeError, match="over the budget"):
        kr.check(fat)
    spilled = {n: dict(r, scratch=8, spill=2, scratch_ops=3) if "tile_partition_kernel<4>" in n else r for n, r in ks.items()}
    with pytest.raises(RuntimeError, match="scratch"):
        kr.check(spilled)
