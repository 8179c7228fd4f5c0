"""CPU tests of the host logic and of the C-ABI library (load + exported symbols; no compute without a GPU)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "benlsip_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import benlsip_jl_amd as bh
    lib = bh.load()
    names = _header_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), "include/benlsip_hip.h declares %s but the library does not export it" % name
    assert set(bh._lib.EXPORTS) == set(names)
    assert lib.bh_strerror(0) == b"ok"
    assert b"HIP" in lib.bh_strerror(-3)


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: without a GPU every compute entry point raises."""
    import benlsip_jl_amd as bh
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bh.BenlsipHipError):
        bh.init(0)
    lib = bh.load()
    assert lib.bh_synchronize() == -2       # BH_ERR_NOT_INIT


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "benlsip.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "benlsip_ref" not in text and "oracle/" not in text and "oracle." not in text, f


def test_pack_bitvector_matches_julia_chunks():
    import benlsip_jl_amd as bh
    f = np.zeros(130, dtype=bool)
    f[[0, 63, 64, 129]] = True
    ch = bh.pack_bitvector(f)
    assert ch.dtype == np.uint64 and ch.shape == (3,)
    assert int(ch[0]) == (1 | (1 << 63)) and int(ch[1]) == 1 and int(ch[2]) == 2
    assert bh.pack_bitvector(np.zeros(0, dtype=bool)).shape == (1,)


def test_row_shard_partitions_rows():
    import benlsip_jl_amd as bh
    for d, G in [(524288, 8), (65536, 4), (10, 3), (7, 8)]:
        ranges = [bh.row_shard(d, k, G) for k in range(G)]
        assert ranges[0][0] == 0 and ranges[-1][1] == d
        for (a, b), (c, e) in zip(ranges, ranges[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    assert bh.row_shard(524288, 3, 8) == (3 * 65536, 4 * 65536)
    with pytest.raises(ValueError):
        bh.row_shard(10, 3, 3)


def test_cg_status_codes_match_header():
    import benlsip_jl_amd as bh
    text = open(os.path.join(ROOT, "include", "benlsip_hip.h")).read()
    for name, member in [("BH_CG_SOLVED", "solved"), ("BH_CG_BOUND_HIT", "bound_hit"), ("BH_CG_NEGATIVE_CURVATURE", "negative_curvature"),
                         ("BH_CG_MAX_ITER_REACHED", "max_iter_reached"), ("BH_CG_NONE", "none")]:
        val = int(re.search(r"#define\s+%s\s+(-?\d+)" % name, text).group(1))
        assert int(bh.CGStatus[member]) == val


def test_graft_entry_build_compiles():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    import benlsip_jl_amd as bh
    assert os.path.exists(bh.library_path())


def test_options_and_argument_checks_without_gpu():
    """Entry points that need no device validate their arguments and report through the error-code convention."""
    import ctypes as C
    import benlsip_jl_amd as bh
    lib = bh.load()
    assert lib.bh_set_option(b"pcg_batch", 4) == 0
    assert lib.bh_set_option(b"no_such_option", 1) == -1 and b"no_such_option" in lib.bh_last_error_detail()
    assert lib.bh_set_option(None, 1) == -1
    h = C.c_void_p()
    assert lib.bh_hess_create(C.byref(h), None, 4, 3, 4, None, 0, 1, 1.0) == -2          # BH_ERR_NOT_INIT before bh_init
    assert lib.bh_hess_destroy(None) == 0 and lib.bh_proj_destroy(None) == 0          # destroying NULL is a no-op
    r, n = C.c_int32(-1), C.c_int32(-1)
    assert lib.bh_comm_info(C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (0, 1)
    for code in range(0, -9, -1):
        assert len(lib.bh_strerror(code)) > 0


def test_every_option_key_is_documented_in_the_header():
    """Every key bh_set_option accepts (csrc/bh_api.hip) appears, quoted, in the option list of include/benlsip_hip.h — and the
    header documents no key the library does not know."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "benlsip.jl_amd", "csrc", "bh_api.hip")).read()
    hdr = open(os.path.join(root, "include", "benlsip_hip.h")).read()
    accepted = set(re.findall(r'!strcmp\(key, "([a-z_]+)"\)', src))
    assert len(accepted) >= 15
    i = hdr.index("int32_t bh_set_option(")
    documented = set(re.findall(r'^ \*   "([a-z_]+)"', hdr[hdr.rindex("/*", 0, i):i], flags=re.M))
    assert accepted - documented == set(), "options without a line in the header: %s" % sorted(accepted - documented)
    assert documented - accepted == set(), "header documents unknown options: %s" % sorted(documented - accepted)


def test_tools_and_product_never_touch_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.  No script
    under tools/ and no file of the product package mentions the oracle modules."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for pat in ("tools/*.py", "tools/*.sh", "tools/*/*.sh", "tools/*/*.py", "benlsip.jl_amd/*.py", "benlsip.jl_amd/csrc/*", "julia/*.jl", "examples/*"):
        for f in glob.glob(os.path.join(root, pat)):
            if not os.path.isfile(f):
                continue
            text = open(f, errors="replace").read()
            if "benlsip_ref" in text or "benlsip_oracle" in text or "oracle/" in text or 'join(ROOT, "oracle")' in text:
                offenders.append(os.path.relpath(f, root))
    assert offenders == [], offenders


def test_bench_self_launch_never_touches_hip_in_the_parent(tmp_path):
    """`python bench.py --gpus N` typed without a launcher starts its ranks as child processes; the launching process must
    not have torch imported or libamdhip64 mapped (a process that has initialised the GPU must never be forked/replaced on
    the GPU boxes), relays rank 0's stdout only, and returns the worst child exit code."""
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import os, sys, json\n"
        "r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "print(json.dumps({'rank': r, 'world': w, 'argv': sys.argv[1:]}), flush=True)\n"
        "print('note from rank %d' % r, file=sys.stderr, flush=True)\n"
        "sys.exit(int(os.environ.get('STUB_FAIL_RANK', '-1')) == r and 7 or 0)\n")
    driver = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import bench\n"
        "rc = bench.self_launch(3, ['--gpus', '3', '--steps', '2'], script=%r)\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert 'libamdhip64' not in maps and 'libbenlsip_hip' not in maps and 'librccl' not in maps, 'GPU runtime mapped in the launcher'\n"
        "assert 'torch' not in sys.modules and 'benlsip_jl_amd' not in sys.modules\n"
        "sys.exit(rc)\n" % (ROOT, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, "-c", driver], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    lines = [json.loads(ln) for ln in ok.stdout.splitlines()]
    assert lines == [{"rank": 0, "world": 3, "argv": ["--gpus", "3", "--steps", "2"]}]          # rank 0's line only
    assert all("[rank %d] note from rank %d" % (r, r) in ok.stderr for r in range(3))
    bad = subprocess.run([sys.executable, "-c", driver], env=dict(env, STUB_FAIL_RANK="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 7 and "rank 2 exited with code 7" in bad.stderr


def test_bench_gpus_n_as_typed_becomes_the_launcher():
    """The command line of the driver's scaling leg, typed without torch.distributed.run, must reach the ranks (here, without
    a GPU, every rank fails loudly in bh.init and the launcher reports a non-zero code instead of refusing to start)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(BH_BENCH_REHEARSAL="1", BH_BENCH_PEER_GRACE_S="5")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_multirank_gpu.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr
    assert "[rank 0]" in r.stderr and "[rank 1]" in r.stderr and "exited with code" in r.stderr


def test_stats_struct_matches_the_header(tmp_path):
    """bh_stats_t as the ctypes mirror declares it against the header, field by field: a small C program compiled against
    include/benlsip_hip.h prints sizeof and every offsetof; names, order, offsets and total size must agree (a field added on one
    side only would shift everything behind it silently)."""
    import ctypes as C
    import re
    import subprocess
    from benlsip_jl_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "benlsip_hip.h")).read()
    body = re.search(r"typedef struct bh_stats_t \{(.*?)\} bh_stats_t;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(int64_t|double)\s+(.*)", decl, re.S)
        assert m, decl
        names += [x.strip() for x in m.group(2).split(",")]
    assert names == [f for f, _ in _lib.bh_stats_t._fields_], (names, [f for f, _ in _lib.bh_stats_t._fields_])
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "benlsip_hip.h"\nint main(void) {\n  printf("%zu\\n", sizeof(bh_stats_t));\n'
                   + "".join('  printf("%%zu\\n", offsetof(bh_stats_t, %s));\n' % f for f in names) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert out[0] == C.sizeof(_lib.bh_stats_t)
    assert out[1:] == [getattr(_lib.bh_stats_t, f).offset for f in names]


def test_python_mirror_keeps_the_reference_argument_order():
    """benlsip.jl_amd/operators.py mirrors the reference's operators for the path: same names, same positional parameters in the same
    order as tests/golden/reference_signatures.json records them — except the augmented factor `chol_aat`, which the device's
    reduced projection form does not take (cauchy_step, inner_step), and linesearch's `fix_bounds`, read from `lincons`."""
    import inspect
    import json
    import benlsip_jl_amd as bh
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "reference_signatures.json")))
    by = {}
    for f in ("basic_tralcnlss.jl", "polyhedral_constraints.jl"):
        for s in ref[f]:
            by.setdefault(s["name"].split(".")[-1].rstrip("!"), []).append(s["positional"])
    for name in ("projected_cg", "minor_iterate", "cauchy_step", "inner_step", "vthv", "projection", "linesearch", "factor_to_boundary"):
        fn = getattr(bh, name)
        mine = [p.name for p in inspect.signature(fn).parameters.values()
                if p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        wanted = [[a if a != "fix_bounds" else "lincons" for a in sig if a != "chol_aat"] for sig in by[name]]
        assert mine in wanted, (name, mine, wanted)


def test_citations_point_into_the_reference():
    """`include/benlsip_hip.h` cites, for every entry point, the reference lines it replaces.  Against the interface fixture
    (tests/golden/reference_signatures.json: first line of every reference function, length of each file): a citation of the form
    `name(args) — src/file.jl:A-B` names a function of the reference whose definition starts inside [A, B], and no citation anywhere
    in the header, the design documents or the oracle points past the end of its file."""
    import glob
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "reference_signatures.json")))
    starts = {}
    for f in ("basic_tralcnlss.jl", "polyhedral_constraints.jl"):
        for s in ref[f]:
            starts.setdefault((f, s["name"].split(".")[-1].lstrip(":")), []).append(s["line"])
    header = open(os.path.join(root, "include", "benlsip_hip.h")).read()
    n_checked = 0
    for m in re.finditer(r"([A-Za-z_][\w.:*]*!?)\s*\([^)\n]*\)[^\n—]*— src/(\w+\.jl):(\d+)(?:-(\d+))?", header):
        name, f, a = m.group(1).split(".")[-1].lstrip(":"), m.group(2), int(m.group(3))
        b = int(m.group(4) or a)
        if (f, name) in starts:
            assert any(a <= s <= b for s in starts[(f, name)]), (name, f, a, b, starts[(f, name)])
            n_checked += 1
    assert n_checked >= 10
    files = [os.path.join(root, "include", "benlsip_hip.h"), os.path.join(root, "DESIGN.md"), os.path.join(root, "INTEGRATION.md"),
             os.path.join(root, "julia", "BEnlsipHIP.jl")] + glob.glob(os.path.join(root, "oracle", "*.py")) + glob.glob(os.path.join(root, "oracle", "*.c"))
    for path in files:
        for f, a, b in re.findall(r"(basic_tralcnlss\.jl|polyhedral_constraints\.jl):(\d+)(?:-(\d+))?", open(path).read()):
            assert int(b or a) <= ref["n_lines"][f] and int(a) >= 1 and int(a) <= int(b or a), (os.path.relpath(path, root), f, a, b)
