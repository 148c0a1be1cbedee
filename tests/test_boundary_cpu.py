"""CPU: the drop-in boundary -- constructor, state_dict contract, init parity,
flat parameter storage, loud failure without a GPU, C-ABI symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd import _lib


def test_state_dict_keys_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "state_dict_census.npz"))
    for name, J, norm in (("nyu_instance", 14, "instance"), ("nyu_batch", 14, "batch"), ("msra_instance", 21, "instance")):
        m = PixelwiseRegression(J, stage=2, label_size=64, features=128, level=4, kernel_size=3, norm_method=norm,
                                heatmap_method="softmax")
        sd = m.state_dict()
        assert list(sd.keys()) == list(g[name + "_keys"])
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g[name + "_shapes"])


def test_parameter_counts():
    # measured parameter counts of the reference
    for J, n in ((16, 3298080), (14, 3288340), (21, 3322430), (42, 3424700)):
        m = PixelwiseRegression(J, stage=2, label_size=64, features=128, level=4, norm_method="instance")
        assert sum(p.numel() for p in m.parameters()) == n


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_same_seed_same_init_as_reference(golden_dir, norm):
    g = np.load(os.path.join(golden_dir, "init_seed1234.npz"))
    torch.manual_seed(1234)
    m = PixelwiseRegression(4, stage=2, label_size=16, features=32, level=2, kernel_size=3, norm_method=norm,
                            heatmap_method="softmax")
    sd = m.state_dict()
    assert list(sd.keys()) == list(g[norm + "_keys"])
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g[norm + "_sum"], rtol=0, atol=1e-9)
    for k in g.files:
        if k.startswith(norm + "_t_"):
            assert np.array_equal(sd[k[len(norm) + 3:]].numpy(), g[k]), k


def test_unknown_norm_raises_like_reference():
    with pytest.raises(UnboundLocalError):
        PixelwiseRegression(4, norm_method="layer")


def test_sum_method_has_no_w():
    m = PixelwiseRegression(4, stage=1, label_size=16, features=32, level=1, heatmap_method="sum")
    assert not any(k.endswith(".w") for k in m.state_dict())


def test_flat_storage_and_load_state_dict(golden_dir):
    from weights_util import fill_state_dict
    m = PixelwiseRegression(4, stage=2, label_size=16, features=32, level=2, norm_method="batch")
    flat = m.flat_parameters()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    sd = fill_state_dict(m.state_dict(), seed=3)
    m.load_state_dict(sd, strict=True)
    assert m.flat_parameters() is flat
    off = 0
    for name, p in m.named_parameters():
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        assert torch.equal(p.detach(), sd[name])
        off += p.numel()
    # round trip through the reference's checkpoint format (utils.py:302-314)
    m2 = PixelwiseRegression(4, stage=2, label_size=16, features=32, level=2, norm_method="batch")
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.flat_parameters(), flat)
    m3 = m2.double().float()     # _apply re-flattens
    assert m3.flat_parameters().numel() == flat.numel()
    assert torch.equal(m3.flat_parameters(), flat)


def test_cpu_forward_fails_loudly():
    m = PixelwiseRegression(4, stage=1, label_size=16, features=32, level=1, norm_method="instance")
    x = torch.zeros(1, 1, 32, 32)
    with pytest.raises(_lib.PwrError):
        m(x, torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))


def test_c_abi_exports_every_declared_symbol():
    """Every function declared in include/pwr.h is exported by the built library and bound in _lib."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "pwr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pwr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(l, name), "libpwr_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().pwr_abi_version() == _lib.ABI_VERSION


def test_checkpoint_format_matches_reference(tmp_path):
    """utils.py:302-314: {"state_dict","seed","model_param"}; a checkpoint in that format round-trips with strict=True."""
    from pixelwiseregression_amd import save_model, load_model
    from weights_util import fill_state_dict
    kw = dict(stage=2, label_size=16, features=32, level=2, kernel_size=3, norm_method="batch", heatmap_method="softmax")
    m = PixelwiseRegression(4, **kw)
    m.load_state_dict(fill_state_dict(m.state_dict(), seed=5))
    path = str(tmp_path / "MSRA_default_final.pt")
    save_model(m, path, seed=0, model_param=kw)
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"state_dict", "seed", "model_param"}
    assert list(raw["state_dict"].keys()) == list(m.state_dict().keys())
    m2 = PixelwiseRegression(4, **raw["model_param"])
    seed, param = load_model(m2, path, eval_mode=True)
    assert seed == 0 and param == kw and not m2.training
    for (k, a), (_, b_) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b_), k


def test_metric_tail_matches_reference_golden(golden_dir):
    """recover_uvd / uvd2xyz / mean joint error (utils.py:332-337, datasets.py:100-111, train.py:276) vs reference outputs."""
    from pixelwiseregression_amd import recover_uvd, uvd2xyz, mean_joint_error, INTRINSICS
    g = np.load(os.path.join(golden_dir, "metric.npz"))
    for name, (fx, fy, hu, hv) in INTRINSICS.items():
        uvd = torch.from_numpy(g[name + "_uvd"].copy())
        rec = recover_uvd(uvd, torch.from_numpy(g[name + "_box"]), torch.from_numpy(g[name + "_com"]), torch.from_numpy(g[name + "_cube"]))
        np.testing.assert_allclose(rec.numpy(), g[name + "_rec"], rtol=1e-6, atol=1e-4)
        xyz = uvd2xyz(rec.numpy(), fx, fy, hu, hv)
        np.testing.assert_allclose(xyz, g[name + "_xyz"], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(mean_joint_error(xyz, g[name + "_xyz_gt"]), g[name + "_err"], rtol=1e-5, atol=1e-3)


def test_engine_plan_builds_without_a_gpu_and_names_its_arena():
    """The launch plan is built on the host (no HIP call): the parameter table of the module must be consumed exactly and the plan has
    stage + 1 backward segments."""
    import ctypes
    import torch
    from pixelwiseregression_amd import PixelwiseRegression, _lib
    l = _lib.lib()
    for joints, P, B in ((14, 64, 32), (21, 64, 8), (4, 16, 2)):
        m = PixelwiseRegression(joints, stage=2, label_size=P, features=128, level=4, norm_method="instance")
        cfg = (ctypes.c_int * 8)(joints, 2, P, 128, 4, 3, 0, 0)
        offs = [o for (o, _) in m._offsets.values()]
        nums = [int(torch.Size(s).numel()) for (_, s) in m._offsets.values()]
        n = len(offs)
        h = l.pwr_engine_create(cfg, B, 1, 1, (ctypes.c_longlong * n)(*offs), (ctypes.c_longlong * n)(*nums), n, None, 0)
        assert h, l.pwr_last_error()
        try:
            assert l.pwr_engine_num_segments(h) == 3
            arena = l.pwr_engine_arena_bytes(h)
            assert arena > 2 * B * (2 * P) ** 2 * 128 * 2 and l.pwr_engine_num_launch_ops(h, 0) > 50     # more than the widest stem tensor + its gradient
        finally:
            l.pwr_engine_destroy(h)
    # a parameter table that does not match the architecture is refused
    bad = (ctypes.c_longlong * n)(*([nums[0] + 1] + nums[1:]))
    assert not l.pwr_engine_create(cfg, 2, 1, 1, (ctypes.c_longlong * n)(*offs), bad, n, None, 0)


def test_reference_written_checkpoint_loads_strict(golden_dir):
    """A .pt written by the reference's own utils.save_model (utils.py:302-307; generated by oracle/gen_golden.py from a
    reference module) loads into the build's module with strict=True through load_model, key for key and bit for bit."""
    from pixelwiseregression_amd import load_model
    path = os.path.join(golden_dir, "reference_checkpoint.pt")
    raw = torch.load(path, map_location="cpu")
    m = PixelwiseRegression(4, **raw["model_param"])
    seed, param = load_model(m, path, eval_mode=True)
    assert seed == 4321 and param == raw["model_param"] and not m.training
    sd = m.state_dict()
    assert list(sd.keys()) == list(raw["state_dict"].keys())
    for k, v in raw["state_dict"].items():
        assert torch.equal(sd[k].cpu(), v), k
    assert int(sd["conv.1.num_batches_tracked"]) == 17


def test_plan_cache_is_bounded():
    """engine._get_plan keeps at most MAX_PLANS plans per module (a ragged last batch must not pin a second arena forever).
    Exercised with a stub plan class: no GPU needed."""
    from pixelwiseregression_amd import engine

    class FakePlan:
        def __init__(self, model, B, dtype, need_grad):
            self.arena = torch.empty(B * 1000, dtype=torch.uint8)
    real, real_budget = engine._Plan, engine._plan_budget_bytes
    engine._Plan, engine._plan_budget_bytes = FakePlan, (lambda dev: 10 ** 12)
    try:
        m = PixelwiseRegression(4, stage=1, label_size=16, features=32, level=1, norm_method="instance")
        sync = torch.cuda.synchronize
        synced = []
        torch.cuda.synchronize = lambda dev=None: synced.append(dev)       # eviction synchronises the whole device (any stream)
        try:
            for B in (1, 2, 3, 4, 5, 6):
                engine._get_plan(m, B, 0, False)
            assert len(m._engine) == engine.MAX_PLANS and (1, 0, False) not in m._engine and (6, 0, False) in m._engine
            engine._get_plan(m, 3, 0, False)                       # touching a plan makes it the most recent
            engine._get_plan(m, 7, 0, False)
            assert (3, 0, False) in m._engine and (4, 0, False) not in m._engine
            engine._plan_budget_bytes = lambda dev: 9000            # byte budget: only what fits stays
            engine._get_plan(m, 8, 0, False)
            assert list(m._engine) == [(8, 0, False)] and len(synced) >= 7
        finally:
            torch.cuda.synchronize = sync
    finally:
        engine._Plan, engine._plan_budget_bytes = real, real_budget


def test_check_flat_detects_a_detached_middle_parameter():
    m = PixelwiseRegression(4, stage=1, label_size=16, features=32, level=1, norm_method="instance")
    flat0 = m.flat_parameters()
    mid = m._param_list[len(m._param_list) // 2]
    mid.data = mid.data.clone() * 0 + 3.0           # a manual .data assignment on a middle parameter
    m._check_flat()
    assert m.flat_parameters() is not flat0, "re-flattened"
    o = m._byte_offsets[len(m._param_list) // 2] // 4
    assert float(m.flat_parameters()[o]) == 3.0     # ... and the new values are in the flat buffer
    assert all(p.data_ptr() == m.flat_parameters().data_ptr() + b for p, b in zip(m._param_list, m._byte_offsets))


def test_no_cross_half_packed_f32_in_shipped_code_objects():
    """Static regression test of the round-2 reproducibility fix: the gfx950 code objects inside
    libpwr_hip.so contain no packed f32 instruction whose low result reads a source's high register (op_sel:[..1..]) -- the
    compiler-generated (SLP) form caught producing a wrong addend in lanes 48-63 beside MFMA kernels of another stream."""
    import pytest
    from pixelwiseregression_amd import _lib, build, codeobj_scan as mod
    assert "-fno-slp-vectorize" in build.FLAGS
    if not mod.available() or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs llvm-objdump / llvm-objcopy of the ROCm toolchain and a built libpwr_hip.so")
    r = mod.scan(_lib.LIB_PATH)
    assert r["functions"] > 100 and r["instructions"] > 100000, r       # the scan really saw the library's kernels
    assert r["packed_f32_cross_half_op_sel"] == 0, r
    assert r["async_lds_hazards"] == 0, r["async_lds_examples"]         # no use of an inline-asm LDS read before its wait (conv_wgrad_dma / _ws)


def test_weight_stationary_conv_keeps_everything_in_registers():
    """csrc/conv_wstat.hip holds 288 weight registers, 64 accumulators and its staging / epilogue state in the 512 registers of a wave, with
    every register index a compile-time constant of its fully unrolled K loops.  If an edit (or a toolchain) breaks that -- the K loop no
    longer unrolls, the allocator spills -- the kernel still computes the right thing, slowly, through scratch memory: refuse such a build."""
    import re
    import subprocess
    import tempfile
    import pytest
    from pixelwiseregression_amd import _lib, codeobj_scan as cs
    if not cs.available() or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs the LLVM tools of the ROCm toolchain and a built libpwr_hip.so")
    seen = 0
    for triple, blob in cs.code_objects(_lib.LIB_PATH):
        if "gfx950" not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([os.path.join(cs.LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S*conv3x3_wstat\S*).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
            name, scratch, vgpr, spill = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
            seen += 1
            assert scratch == 0 and spill == 0 and vgpr <= 512, (name, scratch, vgpr, spill)
    assert seen >= 5, seen


def test_shipped_library_carries_a_clean_scan_record():
    """build.build() writes <lib>.scan.json after every link: the sha256 of the library it scanned and the (clean) result.  A product
    library without a matching record -- linked by hand, or built with PWR_ALLOW_UNSCANNED=1 on a toolchain without llvm-objdump -- is
    refused here, wherever the tests run (the record travels with the library; no disassembler is needed to check it)."""
    import json
    import pytest
    from pixelwiseregression_amd import _lib, codeobj_scan as mod
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpwr_hip.so not built")
    assert os.path.exists(mod.stamp_path(_lib.LIB_PATH)), "no scan record beside the library: build it with pixelwiseregression_amd.build"
    rec = json.load(open(mod.stamp_path(_lib.LIB_PATH)))
    assert rec["sha256"] == mod.lib_digest(_lib.LIB_PATH), "the scan record belongs to another build of the library"
    assert rec.get("scanned") is True and rec["packed_f32_cross_half_op_sel"] == 0 and rec["async_lds_hazards"] == 0, rec


def test_shipped_library_has_one_configuration():
    """Round-3 hygiene: the product library reads no experiment switch (no getenv among its undefined symbols: every PWR_* switch is a
    compile-time constant outside the debug build), exports none of the debugging entry points of include/pwr_debug.h, and the
    variant libraries of earlier rounds are gone from the package directory."""
    import glob
    import subprocess
    import pytest
    from pixelwiseregression_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpwr_hip.so not built")
    assert _lib.LIB_PATH.endswith(os.path.join("pixelwiseregression_amd", "libpwr_hip.so")) and "PWR_LIB" not in open(_lib.__file__).read()
    nm = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
    undefined = {l.split()[-1].split("@")[0] for l in nm if " U " in l}
    exported = {l.split()[-1] for l in nm if " T " in l}
    assert "getenv" not in undefined and "secure_getenv" not in undefined
    assert not [e for e in exported if e.startswith("pwr_debug")] and "pwr_engine_set_join" not in exported and "pwr_engine_layout" not in exported
    # ... and no C++ (mangled) debugging setter or debug-only experiment either: pwr::set_debug_*, the nine-tap weight-gradient experiment
    allsyms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    dem = subprocess.run(["c++filt"], input=allsyms, capture_output=True, text=True, check=True).stdout
    assert not [l for l in dem.splitlines() if "debug" in l.lower() or "wgrad9w" in l], [l for l in dem.splitlines() if "debug" in l.lower() or "wgrad9w" in l][:5]
    assert not os.path.exists(os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "conv_wgrad_ws9.hip"))
    assert set(_lib.SIGNATURES) <= exported
    assert glob.glob(os.path.join(os.path.dirname(_lib.LIB_PATH), "*.so")) == [_lib.LIB_PATH]


def test_one_pixel_innermost_map_fails_like_the_reference():
    """label_size / 2^(level+1) == 1: the reference's InstanceNorm2d raises ValueError in F.instance_norm (train and eval), its
    BatchNorm2d only when training on a single sample; same exceptions here, before anything touches the GPU."""
    import pytest
    from pixelwiseregression_amd import PixelwiseRegression
    x = torch.zeros(2, 1, 64, 64), torch.zeros(2, 1, 32, 32), torch.ones(2, 1, 32, 32)
    m = PixelwiseRegression(3, stage=1, label_size=32, features=32, level=4, norm_method="instance")
    for mode in (m.train, m.eval):
        mode()
        with pytest.raises(ValueError, match="Expected more than 1 spatial element"):
            m(*x)
    mb = PixelwiseRegression(3, stage=1, label_size=32, features=32, level=4, norm_method="batch").train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        mb(*(t[:1] for t in x))


def test_input_gradients_are_refused_loudly():
    """The engine's backward stops at the parameters; an input that requires grad must not silently get none."""
    import pytest
    from pixelwiseregression_amd import PixelwiseRegression
    m = PixelwiseRegression(3, stage=1, label_size=16, features=32, level=1, norm_method="instance")
    img = torch.zeros(1, 1, 32, 32, requires_grad=True)
    with pytest.raises(NotImplementedError, match="gradients with respect to img"):
        m(img, torch.zeros(1, 1, 16, 16), torch.ones(1, 1, 16, 16))


def test_package_reads_no_experiment_variable():
    """The Python package, like the library, has ONE configuration: no module of pixelwiseregression_amd reads a PWR_* environment
    variable (round 3 left two: the default precision and the plan-cache budget; they are a constructor default and a module attribute
    now).  HIPCC (the compiler to build with) and PWR_ALLOW_UNSCANNED (an explicit build opt-out that the tests flag) are build inputs."""
    import glob
    import re
    from pixelwiseregression_amd import _lib
    pkg = os.path.dirname(_lib.__file__)
    for f in glob.glob(os.path.join(pkg, "*.py")):
        for m in re.finditer(r"environ[^\n]*?[\"'](PWR_[A-Z0-9_]+)", open(f).read()):
            assert m.group(1) == "PWR_ALLOW_UNSCANNED" and os.path.basename(f) == "build.py", (f, m.group(1))
