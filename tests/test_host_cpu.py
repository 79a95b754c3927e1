"""CPU-side checks of the product: C-ABI library loads and exports every declared symbol, module
surface / state_dict layout, host logic.  No compute kernels are called (no GPU here)."""
import json
import os

import numpy as np
import pytest
import torch

from common import GOLD


def test_library_exports_every_declared_symbol():
    from cofii2p_amd import _lib, build

    build.build()
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 30
    assert set(declared) == set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cofi_abi_version() == _lib.ABI_VERSION and lib.cofi_target_arch() == b"gfx950"
    # the tuning / test hooks are declared too (include/cofi_hip_tune.h), and the two headers together are EVERYTHING the library exports
    tune = _lib.header_symbols(_lib.TUNE_HEADER_PATH)
    assert set(tune) == set(_lib.TUNE_SIGNATURES) and not set(tune) & set(declared)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("cofi_")}
    assert exported == set(declared) | set(tune), sorted(exported ^ (set(declared) | set(tune)))
    # pure host-side queries are callable without a GPU
    assert lib.cofi_gemm_f32_workspace(1280, 512, 7680) > 0
    assert lib.cofi_gemm_f32_workspace(20480, 128, 64) == 0
    assert lib.cofi_group_stats_workspace(1280, 2048, 32, 1) == 20 * 32 * 2 * 8
    assert lib.cofi_group_stats_workspace(1280, 2048, 32, 16) == 2 * 16 * 32 * 2 * 8  # 16 frames of 80 rows: 2 slabs each


def test_descriptor_structs_mirror_the_header():
    """the ctypes mirrors of the two descriptor structs of the C ABI declare the header's fields, in the header's order, with matching
    C types (a drifted field would silently shift every later argument), and the argument counts of the binding table equal the header's
    parameter counts for every entry point"""
    import ctypes
    import re

    from cofii2p_amd import _lib

    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    ctype_of = {"int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t}

    def fields_of(struct_name):
        body = re.search(r"typedef struct[^{}]*\{([^{}]*)\}\s*%s;" % struct_name, text).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            base = re.match(r"(const\s+)?(\w+)", decl).group(2)
            for part in decl[re.match(r"(const\s+)?\w+", decl).end():].split(","):
                part = part.strip()
                ptr = part.startswith("*")
                m = re.match(r"\*?\s*(\w+)(\[(\d+)\])?", part)
                t = ctypes.c_void_p if ptr else ctype_of[base]
                out.append((m.group(1), t * int(m.group(3)) if m.group(3) else t))
        return out

    for cname, mirror in (("cofi_norm_desc_t", _lib.NormDesc), ("cofi_loftr_tail_desc_t", _lib.TailDesc)):
        want, got = fields_of(cname), list(mirror._fields_)
        assert [n for n, _ in want] == [n for n, _ in got], cname
        for (n, tw), (_, tg) in zip(want, got):
            assert ctypes.sizeof(tw) == ctypes.sizeof(tg) and (tw is tg or issubclass(tg, ctypes.Array) == issubclass(tw, ctypes.Array)), (cname, n)
    for name, (_res, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_code_object_targets_gfx950_only():
    from cofii2p_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"gfx90a" not in blob


def test_state_dict_layout_matches_reference_dump():
    from cofii2p_amd.network import CoFiI2P

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"

    m = CoFiI2P(Opt())
    ref = json.load(open(os.path.join(GOLD, "state_dict_spec.json")))
    sd = m.state_dict()
    assert [k for k, _, _ in ref] == list(sd.keys())
    for k, shape, dtype in ref:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype, k
    assert sum(p.numel() for p in m.parameters()) == 51588744
    # strict loading of a foreign checkpoint with the reference's keys, and pack invalidation
    other = {k: torch.zeros_like(v) for k, v in sd.items()}
    m._packed = {"stale": None}
    m.load_state_dict(other, strict=True)
    assert m._packed is None
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(other.items())[:-1]}, strict=True)


@pytest.mark.parametrize("norm,n_params", [("bn", 51588744), ("ln", 51588744)])
def test_state_dict_layout_of_the_other_norm_configurations(norm, n_params):
    """opt.norm = 'bn' / 'ln' (get_norm(), model/kpconv/modules.py:51-60): key order, shapes and dtypes of the reference's own
    state_dict for that option (tests/golden/state_dict_spec_<norm>.json, dumped from the reference)."""
    from cofii2p_amd.network import CoFiI2P

    class Opt:
        img_H, img_W, img_fine_resolution_scale = 160, 512, 32

    Opt.norm = norm
    m = CoFiI2P(Opt())
    ref = json.load(open(os.path.join(GOLD, "state_dict_spec_%s.json" % norm)))
    sd = m.state_dict()
    assert [k for k, _, _ in ref] == list(sd.keys())
    for k, shape, dtype in ref:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype, k
    assert sum(p.numel() for p in m.parameters()) == n_params
    Opt.norm = "in"
    with pytest.raises(ValueError):
        CoFiI2P(Opt())


def test_forward_has_no_cpu_path():
    from cofii2p_amd._lib import CofiError
    from cofii2p_amd.network import CoFiI2P

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"

    with pytest.raises(CofiError):
        CoFiI2P(Opt())({"points": [], "neighbors": [], "subsampling": [], "upsampling": [], "feats": torch.zeros(1, 4)},
                       torch.zeros(1, 3, 160, 512), None, None, None, "test")


def test_threshold_sequence_and_frequencies():
    import cofi_oracle as O
    from cofii2p_amd import ops
    from cofii2p_amd.network import score_thresholds

    t = score_thresholds()
    assert t.dtype == np.float32 and t[0] == np.float32(0.9) and len(t) == 64
    assert np.array_equal(t, np.asarray(O.score_thresholds(), dtype=np.float64).astype(np.float32))
    f = ops.sine_frequencies(3)
    assert f.shape == (42,) and f[0] == 1.0 and f[1] == 1.0
    assert ops.sine_frequencies(2).shape == (64,)


def test_synthetic_frame_is_deterministic_and_lattice_like():
    from cofii2p_amd.synth import make_frame, subsample_indices

    a, b = make_frame(3, 2048), make_frame(3, 2048)
    assert np.array_equal(a.points, b.points) and np.array_equal(a.img, b.img) and a.points.shape == (2048, 3)
    assert a.feats.shape == (2048, 4) and np.allclose(np.linalg.norm(a.feats[:, 1:], axis=1), 1, atol=1e-5)
    s = subsample_indices(2048, 5, 1)
    assert [len(x) for x in s] == [1024, 512, 256, 128]


def test_spec_tables():
    from cofii2p_amd import spec

    assert [b.mid for b in spec.ENCODER][:3] == [64, 32, 32]
    assert abs(spec.ENCODER[-1].sigma - 3.2) < 1e-6 and spec.stage_sizes(20480) == [20480, 10240, 5120, 2560, 1280]
    kp = spec.synth_state_dict()["pc_encoder.encoder3_2.KPConv.kernel_points"]
    assert kp.shape == (15, 3) and np.allclose(kp[0], 0) and abs(np.linalg.norm(kp[1]) - 0.66 * 0.425 * 4) < 1e-5


def test_integration_md_ctypes_example_matches_the_abi():
    """the binding stub INTEGRATION.md shows a maintainer must declare as many arguments as the header / binding table"""
    import re

    from cofii2p_amd import _lib

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    found = re.findall(r"_lib\.(cofi_\w+)\.argtypes = \[(.*?)\]", text, re.S)
    assert len(found) >= 3
    for name, body in found:
        n = len([x for x in body.replace("\n", " ").split(",") if x.strip()])
        assert n == len(_lib.SIGNATURES[name][1]), name


def test_model_import_path_shim_and_training_guard():
    """`from model.network import CoFiI2P` (the reference's import path) resolves to this implementation; mode='train' with
    autograd enabled raises instead of returning tensors without a graph"""
    import model.network as shim
    from model.kpconv.preprocess_data import precompute_point_cloud_cuda, precompute_point_cloud_stack_mode

    from cofii2p_amd import network, preprocess

    # the shim's class IS the implementation, specialised only in its default arithmetic (exact fp32: test_arithmetic_is_an_explicit_option...)
    assert issubclass(shim.CoFiI2P, network.CoFiI2P) and shim.point2node is network.point2node and issubclass(shim.CoFiI2P_wrapper, network.CoFiI2P_wrapper)
    assert set(vars(shim.CoFiI2P)) - {'__module__', '__doc__', '__qualname__'} == {'DEFAULT_ARITHMETIC', 'DEFAULT_GRAPHS'}
    assert precompute_point_cloud_stack_mode is preprocess.precompute_point_cloud_stack_mode is precompute_point_cloud_cuda

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"

    m = network.CoFiI2P(Opt())
    from cofii2p_amd import _lib

    with pytest.raises(_lib.CofiError):   # the training path has no CPU form either
        m({"feats": torch.zeros(4, 4)}, torch.zeros(1, 3, 160, 512), None, None, None, "train")
    assert all(p.requires_grad for p in m.parameters())   # learnable, as train.py:163 expects


def test_bench_refuses_more_ranks_than_gpus():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 2 but this node exposes 0 GPU(s)" in (out.stderr + out.stdout)
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "4", "0", "0"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 1 but WORLD_SIZE=4" in (out.stderr + out.stdout)


def test_neighbourhood_operator_shims_and_oracle_properties():
    """model/kpconv/ops/{grid_subsample,radius_search}.py import paths resolve to cofii2p_amd.neighbors; the numpy oracle the GPU tests
    compare against has the properties of the published algorithm (one barycentre per occupied cell, mass preserved; neighbours
    inside the radius, nearest first, shadow index behind them)."""
    import neighbors_oracle as NO
    from cofii2p_amd import neighbors
    from model.kpconv.ops import grid_subsample, radius_search
    from model.kpconv.ops.grid_subsample import grid_subsample as gs2
    from model.kpconv.ops.radius_search import radius_search as rs2

    assert grid_subsample is neighbors.grid_subsample is gs2 and radius_search is neighbors.radius_search is rs2
    g = np.random.default_rng(3)
    p = g.uniform(-4, 4, (4000, 3)).astype(np.float32)
    sp, sl = NO.grid_subsample(p, [2500, 1500], 0.7)
    assert sp.shape[0] == sl.sum() and (sl < np.array([2500, 1500])).all()
    for seg, bary in ((p[:2500], sp[:sl[0]]), (p[2500:], sp[sl[0]:])):
        origin = np.floor(seg.min(0) * (np.float32(1) / np.float32(0.7))) * np.float32(0.7)
        cells = np.floor((seg - origin) / np.float32(0.7)).astype(np.int64)
        uniq, cnt = np.unique(cells[:, [2, 1, 0]], axis=0, return_counts=True)
        assert len(uniq) == len(bary)
        np.testing.assert_allclose((bary.astype(np.float64) * cnt[:, None]).sum(0) / len(seg), seg.astype(np.float64).mean(0), atol=1e-4)
    rows = NO.radius_search(p[:200], p, [120, 80], [2500, 1500], 0.9, 48)
    assert rows.shape[0] == 200 and rows.shape[1] <= 48 and rows.dtype == np.int64
    assert ((rows[:120] < 2500) | (rows[:120] == 4000)).all() and ((rows[120:] >= 2500)).all()
    assert (rows[:120, 0] == np.arange(120)).all()   # a query that is a support point finds itself first
    for r in (0, 57, 150):
        ids = rows[r][rows[r] != 4000]
        d = ((p[ids].astype(np.float64) - p[r].astype(np.float64)) ** 2).sum(1)
        assert (d < 0.81 + 1e-5).all() and (np.diff(d) >= -1e-6).all()


def test_arithmetic_is_an_explicit_option_and_the_shim_is_strict():
    """the reference-named import path must not hand out approximate contractions unasked.  `model.network.CoFiI2P` defaults to
    the fp32-grade arithmetic ("bf16x6": same error against fp64 as exact fp32, tests/test_forward_gpu.py::test_kitti_frame_bf16x6_is_fp32_grade;
    "f32" on request), `cofii2p_amd.network.CoFiI2P` follows the process default unless told, `opt.arithmetic` / the constructor
    argument override both, and the per-forward context restores the process default."""
    from cofii2p_amd import ops
    from cofii2p_amd.network import CoFiI2P as Native
    from model.network import CoFiI2P as Shim, CoFiI2P_wrapper

    class Opt:
        img_H, img_W, img_fine_resolution_scale, norm = 160, 512, 32, "gn"

    assert Shim(Opt()).arithmetic == "bf16x6" and CoFiI2P_wrapper(Opt()).cofii2p.arithmetic == "bf16x6"
    assert Native(Opt()).arithmetic is None and Native(Opt(), arithmetic="f32").arithmetic == "f32" and Shim(Opt(), arithmetic="f32").arithmetic == "f32"
    o = Opt()
    o.arithmetic = "bf16x3"
    assert Shim(o).arithmetic == "bf16x3" and Native(o).arithmetic == "bf16x3"
    with pytest.raises(ValueError):
        Native(Opt(), arithmetic="fp16")
    before = ops.gemm_mode()
    assert before == ops.GEMM_MODE   # no context: the process default (COFI_GEMM)
    with ops.arithmetic("f32"):
        assert ops.gemm_mode() == "f32" and ops.GEMM_MODE == before   # the override is the thread's, the process default stays
        with ops.arithmetic(None):
            assert ops.gemm_mode() == "f32"
        with ops.arithmetic("bf16x6"):
            assert ops.gemm_mode() == "bf16x6"
        assert ops.gemm_mode() == "f32"
        # another thread does not see this thread's override (a loader thread doing inference during a backward pass)
        import threading

        seen = []
        t = threading.Thread(target=lambda: seen.append(ops.gemm_mode()))
        t.start()
        t.join()
        assert seen == [before]
    assert ops.gemm_mode() == before
    assert list(Shim(Opt()).state_dict().keys()) == list(Native(Opt()).state_dict().keys())   # same 430-key layout


def _call_pyramid_in_worker(q):
    import numpy as np

    from model.kpconv.preprocess_data import precompute_point_cloud_stack_mode

    try:
        precompute_point_cloud_stack_mode(np.zeros((3, 64), np.float32), None, None, None, 2)
        q.put("no error")
    except RuntimeError as e:
        q.put(str(e))
    except Exception as e:   # pragma: no cover
        q.put("other: %r" % (e,))


def test_pyramid_shim_refuses_forked_dataloader_workers():
    """the reference calls precompute_point_cloud_stack_mode inside Dataset.__getitem__ (data/kitti.py:292); a forked worker cannot
    initialise HIP - the shim says so instead of hanging"""
    import multiprocessing as mp

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_call_pyramid_in_worker, args=(q,))
    p.start()
    msg = q.get(timeout=60)
    p.join(30)
    assert "forked DataLoader worker" in msg and "FramePreparer" in msg


def test_transposed_table_is_the_csr_transpose_in_pair_order():
    """autograd.TransposedTable (the backward's scatter-free adjoints walk it): for every support row j the ids m * H + h with
    idx[m, h] == j, ascending - the fixed summation order that makes gradients bit-reproducible; shadow entries (== N) are dropped."""
    import torch
    from cofii2p_amd.autograd import TableCache, TransposedTable

    g = torch.Generator().manual_seed(3)
    M, H, N = 37, 5, 11
    idx = torch.randint(0, N + 1, (M, H), generator=g, dtype=torch.int32)
    t = TransposedTable(idx, N)
    flat = idx.reshape(-1)
    assert t.offsets.shape == (N + 1,) and int(t.offsets[0]) == 0 and t.pairs.dtype == torch.int32
    for j in range(N):
        want = torch.nonzero(flat == j).reshape(-1).to(torch.int32)
        got = t.pairs[int(t.offsets[j]):int(t.offsets[j + 1])]
        assert torch.equal(got, want), j
    assert int(t.offsets[N]) == int((flat < N).sum())
    cache = TableCache()
    assert cache.get(idx, N) is cache.get(idx, N) and cache.get(idx, N, first_column=True).H == 1


def test_batch_from_sample_maps_a_loader_sample_the_way_train_py_does():
    """train_step.batch_from_sample = train.py:192-217: squeeze the DataLoader's batch dimension of 1, rename fine_xy_coors -> fine_xy."""
    import torch
    from cofii2p_amd.train_step import batch_from_sample

    K = 6
    sample = {"img": torch.zeros(1, 3, 8, 16), "K_4": torch.eye(3)[None], "P": torch.eye(4)[None],
              "pc_data_dict": {"points": [torch.zeros(1, 32, 3), torch.zeros(1, 16, 3)], "neighbors": [torch.zeros(1, 32, 4, dtype=torch.int64)],
                               "subsampling": [torch.zeros(1, 16, 4, dtype=torch.int64)], "upsampling": [torch.zeros(1, 32, 4, dtype=torch.int64)],
                               "feats": torch.zeros(1, 32, 4)}}
    for k in ("pc_kpt_idx", "pc_outline_idx", "coarse_img_kpt_idx", "fine_pc_inline_index"):
        sample[k] = torch.arange(K)[None]
    sample["fine_center_kpt_coors"] = torch.ones(1, 2, K, dtype=torch.int64)
    sample["fine_xy_coors"] = 2 * torch.ones(1, 2, K, dtype=torch.int64)
    pc, img, batch = batch_from_sample(sample, device="cpu")
    assert img.shape == (1, 3, 8, 16) and pc["points"][0].shape == (32, 3) and pc["neighbors"][0].shape == (32, 4) and pc["feats"].shape == (32, 4)
    assert batch["K_4"].shape == (3, 3) and batch["P"].shape == (4, 4) and batch["pc_kpt_idx"].shape == (K,)
    assert batch["fine_xy"].shape == (2, K) and int(batch["fine_xy"][0, 0]) == 2 and set(batch) == {
        "K_4", "P", "pc_kpt_idx", "pc_outline_idx", "coarse_img_kpt_idx", "fine_center_kpt_coors", "fine_xy", "fine_pc_inline_index"}
    unbatched = {k: (v[0] if torch.is_tensor(v) else v) for k, v in sample.items() if k != "pc_data_dict"}
    unbatched["pc_data_dict"] = {k: ([t[0] for t in v] if isinstance(v, list) else v[0]) for k, v in sample["pc_data_dict"].items()}
    pc2, img2, batch2 = batch_from_sample(unbatched, device="cpu")     # FramePreparer / FrameLoader samples carry no batch dimension
    assert img2.shape == (1, 3, 8, 16) and pc2["points"][1].shape == (16, 3) and batch2["fine_center_kpt_coors"].shape == (2, K)


def test_graphed_train_step_refuses_a_host_side_step_counter():
    import pytest
    import torch
    from cofii2p_amd.train_step import GraphedTrainStep

    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(ValueError):
        GraphedTrainStep(torch.nn.Linear(1, 1), torch.optim.Adam([p], lr=1e-3), None)


def test_attention_arithmetic_follows_the_contraction_arithmetic():
    """ops.attention_arith(): the split-arithmetic attention kernel unless the contractions run on the exact fp32 instruction; COFI_ATTN
    (ops.ATTN_MODE) overrides in both directions; anything else is 'auto'"""
    from cofii2p_amd import ops

    old = ops.GEMM_MODE, ops.ATTN_MODE
    try:
        ops.ATTN_MODE = "auto"
        for gemm, want in (("bf16x6", "bf16x6"), ("bf16x3", "bf16x6"), ("f32", "f32")):
            ops.GEMM_MODE = gemm
            assert ops.attention_arith() == want
        ops.GEMM_MODE = "bf16x6"
        ops.ATTN_MODE = "f32"
        assert ops.attention_arith() == "f32"
        ops.GEMM_MODE, ops.ATTN_MODE = "f32", "bf16x6"
        assert ops.attention_arith() == "bf16x6"
        ops.ATTN_MODE = "something else"
        assert ops.attention_arith() == "f32"   # falls back to the rule
        # the rule reads the CALLING THREAD's arithmetic (ops.arithmetic / CoFiI2P(opt, arithmetic=...)), not the process default: a model
        # built for exact fp32 validation must not run the split-arithmetic attention kernel, and vice versa
        ops.ATTN_MODE = "auto"
        ops.GEMM_MODE = "bf16x6"
        with ops.arithmetic("f32"):
            assert ops.gemm_mode() == "f32" and ops.attention_arith() == "f32"
            with ops.arithmetic("bf16x6"):
                assert ops.attention_arith() == "bf16x6"
            assert ops.attention_arith() == "f32"
        ops.GEMM_MODE = "f32"
        with ops.arithmetic("bf16x6"):
            assert ops.attention_arith() == "bf16x6"
        with ops.arithmetic(None):
            assert ops.attention_arith() == "f32"
    finally:
        ops.GEMM_MODE, ops.ATTN_MODE = old


def test_pack_f16x3_weight_layout_and_scales():
    """ops.pack_f16x3_weight (the COFI_GEMM_W_F16PRE operand, include/cofi_hip.h): groups of four values become {hi0..3, lo0..3} fp16 of w * s in
    place, one power-of-two scale per 128-row panel behind the matrix that puts the panel's largest |w| into [2^11, 2^12); hi + lo restores
    w to 2^-22 of the panel maximum; an all-zero panel keeps scale 1"""
    import torch

    from cofii2p_amd import ops

    g = torch.Generator().manual_seed(4)
    N, K = 300, 96
    w = torch.randn(N, K, generator=g) * torch.tensor([0.01] * 128 + [7.0] * 128 + [0.0] * 44)[:, None]
    buf = ops.pack_f16x3_weight(w)
    assert buf.dtype == torch.float32 and buf.numel() == N * K + 3
    scale = buf[N * K:]
    assert float(scale[2]) == 1.0
    h = buf[:N * K].view(N, K).view(torch.float16).view(N, K // 4, 8)
    hi, lo = h[:, :, :4].reshape(N, K).float(), h[:, :, 4:].reshape(N, K).float()
    for p in range(2):
        rows = slice(128 * p, 128 * (p + 1))
        s = float(scale[p])
        assert s == 2.0 ** round(float(torch.log2(scale[p])))                       # a power of two
        assert 2048.0 <= float(hi[rows].abs().max()) < 4096.0 + 2.0                 # (the maximum may round up to 2^12)
        assert torch.equal(hi[rows], (w[rows] * s).to(torch.float16).float())        # hi = f16(w s), RNE
        assert float(((hi[rows] + lo[rows]) / s - w[rows]).abs().max()) <= 2.0 ** -22 * float(w[rows].abs().max())
    assert float(hi[256:].abs().max()) == 0.0 and float(lo[256:].abs().max()) == 0.0


def test_fragment_order_is_the_mfma_b_operand_order():
    """ops.fragment_order (cofi_loftr_tail_desc_t::w_frag): plane[((T * (K / 16) + s) * 64 + lane) * 8 + i] = W[32 T + (lane & 31)][16 s + 8 (lane >> 5) + i]"""
    import torch

    from cofii2p_amd import ops

    N, K = 64, 48
    w = torch.arange(2 * N * K, dtype=torch.int16).view(2, N, K)
    f = ops.fragment_order(w).reshape(2, -1)
    for (p, T, s, lane, i) in [(0, 0, 0, 0, 0), (1, 1, 2, 37, 5), (0, 1, 1, 63, 7), (1, 0, 2, 31, 3)]:
        assert int(f[p, ((T * (K // 16) + s) * 64 + lane) * 8 + i]) == int(w[p, 32 * T + (lane & 31), 16 * s + 8 * (lane >> 5) + i])
