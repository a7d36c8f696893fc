"""C-ABI checks that need no GPU: the library loads, exports every symbol include/gsage.h
declares, and its host-side legacy stream equals numpy's (the stream the reference consumes)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg


def _declared():
    text = open(os.path.join(ROOT, "include", "gsage.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsage_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    gs = pkg()
    names = _declared()
    assert len(names) >= 15
    lib = ctypes.CDLL(gs._native.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libgsage_hip.so does not export %s" % n
        assert n in gs._native.SIGNATURES, "ctypes binding lacks %s" % n
    assert set(gs._native.SIGNATURES) == set(names)
    assert gs._native.lib().gsage_abi_version() == gs._native.ABI_VERSION == 6


def test_missing_library_fails_loudly(monkeypatch):
    gs = pkg()
    monkeypatch.setattr(gs._native, "_LIB", None)
    monkeypatch.setattr(gs._native, "LIB_PATH", "/nonexistent/libgsage_hip.so")
    with pytest.raises(gs._native.NativeLibraryError):
        gs._native.lib()
    assert not gs._native.available()


def test_bad_arguments_return_einval_without_gpu():
    L = pkg()._native.lib()
    rc = L.gsage_sample_csr_sel(None, None, 10, None, 4, 0, None, None, None, None)
    assert rc == -1 and b"n_samples" in L.gsage_last_error()
    rc = L.gsage_gather_mean(None, 7, 8, None, 4, 1, 8, None, 0, 8, None)
    assert rc == -1
    rc = L.gsage_linear_nt(ctypes.c_void_p(16), 1, 7, None, 0, ctypes.c_void_p(16), 8, None,
                           ctypes.c_void_p(16), 0, 8, 4, 4, 4, 0, 1, 0, 0, 0, None)
    assert rc == -1 and b"lda" in L.gsage_last_error()


def test_command_list_lifecycle_without_gpu():
    """Recording state machine of the command lists (no kernel is recorded, so no GPU needed)."""
    nat = pkg()._native
    L = nat.lib()
    h = ctypes.c_void_p()
    assert L.gsage_cmdlist_end(ctypes.byref(h)) == -1 and b"no recording" in L.gsage_last_error()
    assert L.gsage_cmdlist_begin() == 0
    assert L.gsage_cmdlist_begin() == -1 and b"already recording" in L.gsage_last_error()
    # argument validation still runs while recording, and a rejected call records nothing
    assert L.gsage_gather_mean(None, 7, 8, None, 4, 1, 8, None, 0, 8, None) == -1
    assert L.gsage_cmdlist_end(ctypes.byref(h)) == 0 and h.value
    assert L.gsage_cmdlist_size(h) == 0
    assert L.gsage_cmdlist_replay(h, None) == 0            # empty list: nothing to launch
    L.gsage_cmdlist_destroy(h)
    assert L.gsage_cmdlist_size(None) == -1
    with nat.CommandList.record() as cl:
        pass
    assert len(cl) == 0


@pytest.mark.parametrize("seed", [0, 123, 15129])
def test_host_legacy_stream_equals_numpy(seed):
    L = pkg()._native.lib()
    mt = L.gsage_mt_create(seed)
    try:
        np.random.seed(seed)
        for high, count in ((21657, 1000), (8, 64), (1, 5), (2 ** 31 - 1, 10)):
            out = np.empty(count, dtype=np.int32)
            L.gsage_mt_choice_i32(mt, high, count, out.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(out.astype(np.int64), np.random.choice(high, count))
        perm = np.empty(1030, dtype=np.int64)
        L.gsage_mt_permutation(mt, 1030, perm.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(perm, np.random.permutation(np.arange(1030)))
        L.gsage_mt_seed(mt, seed + 1)
        np.random.seed(seed + 1)
        out = np.empty(7, dtype=np.int32)
        L.gsage_mt_choice_i32(mt, 100, 7, out.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out, np.random.choice(100, 7))
    finally:
        L.gsage_mt_destroy(mt)


def test_product_never_touches_the_oracle_or_the_reference():
    """oracle/ is test infrastructure: nothing under the product package, bench.py's timed path or
    the C sources may import, load or mention it (bench.py's cpu_baseline leg and
    __graft_entry__.smoke() are the two documented exceptions); nothing that runs on the GPU box may
    read /root/reference."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = glob.glob(os.path.join(root, "pytorch-graphsage_amd", "*.py")) + \
        glob.glob(os.path.join(root, "pytorch-graphsage_amd", "engine", "*.py")) + \
        glob.glob(os.path.join(root, "pytorch-graphsage_amd", "csrc", "*")) + \
        glob.glob(os.path.join(root, "include", "*.h"))
    prod = [f for f in prod if os.path.isfile(f) and not f.endswith((".o", ".so"))]
    assert len(prod) > 20
    for f in prod:
        text = open(f, errors="replace").read()
        assert not re.search(r"(^|\W)(import\s+oracle|from\s+oracle|libgsage_oracle|gso_)", text), f
        assert "/root/reference" not in text, f
    bench = open(os.path.join(root, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    body = bench[bench.index("def cpu_baseline("):bench.index("def timed_launches(")]
    assert uses and all(bench.index("def cpu_baseline(") < u < bench.index("def timed_launches(") for u in uses)
    assert "from oracle import" in body
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(root, f)).read() or f == "__graft_entry__.py"


def test_ctypes_structs_mirror_the_header(tmp_path):
    """Every struct that crosses the C ABI by address is declared twice (include/gsage.h and a
    ctypes.Structure): sizes and field offsets must agree -- checked with the C compiler, no GPU."""
    import subprocess
    gs = pkg()
    nat, eng = gs._native, gs.engine
    mirrors = {"gsage_hops_desc": nat.HopsDesc, "gsage_adam_desc": nat.AdamDesc, "gsage_wgrad_desc": nat.WgradDesc, "gsage_row_adam": nat.RowAdamDesc,
               "gsage_tail_gather_desc": nat.TailGatherDesc, "gsage_reduce_desc": eng._ReduceDesc,
               "gsage_prep_desc": eng._PrepDesc}
    lines = []
    for cname, cls in mirrors.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gsage.h"\nint main(void) {\n%s\nreturn 0; }\n'
                   % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in mirrors.items():
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_sampler_role_workgroup_count_is_host_arithmetic(monkeypatch):
    """gsage_mean_tail_mfma_sampler_wgs: how many workgroups the sampler role adds to the seed-level launch, or 0 when
    a workgroup's two frontier buffers (16 bytes per id of the widest hop and seed) would not fit the launch's LDS --
    what the engines size the gather role with (engine/mean.py:_tail_idle_cus)."""
    L = pkg()._native.lib()
    monkeypatch.delenv("GSAGE_TAIL_SMP_WGS", raising=False)
    assert L.gsage_mean_tail_mfma_sampler_wgs(512, 250) == 32          # BASELINE configs[1]: sixteen seeds each
    assert L.gsage_mean_tail_mfma_sampler_wgs(64, 250) == 32           # two seeds each
    assert L.gsage_mean_tail_mfma_sampler_wgs(20, 250) == 20           # never more workgroups than seeds
    assert L.gsage_mean_tail_mfma_sampler_wgs(512, 750) == 0           # configs[4]: 16 x 750 ids twice is 192 KB
    assert L.gsage_mean_tail_mfma_sampler_wgs(0, 250) == 0 and L.gsage_mean_tail_mfma_sampler_wgs(512, 0) == 0
    monkeypatch.setenv("GSAGE_TAIL_SMP_WGS", "7")
    assert L.gsage_mean_tail_mfma_sampler_wgs(200, 250) == 7           # ceil(200 / ceil(200 / 7))
    monkeypatch.setenv("GSAGE_TAIL_SMP_WGS", "64")
    assert L.gsage_mean_tail_mfma_sampler_wgs(512, 750) == 64          # eight seeds each: 96 KB fits
    monkeypatch.setenv("GSAGE_TAIL_SMP_WGS", "100000")                 # out of range: the default
    assert L.gsage_mean_tail_mfma_sampler_wgs(512, 250) == 32
