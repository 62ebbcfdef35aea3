"""C-ABI library: loads without a GPU, exports every symbol include/panfusion_hip.h declares,
validates arguments before launching (no compute here)."""
import ctypes as C
import os
import re

from conftest import ROOT
from panfusion_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "panfusion_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = header_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_plumbing():
    lib = _lib.lib()
    assert lib.pf_version() >= 100
    assert lib.pf_equi_coords(1, 1, None, None) == 1          # PF_ERR_ARG, no launch
    assert b"pf_equi_coords" in lib.pf_last_error_string()
    d = _lib.ConvDesc()
    assert lib.pf_conv_gemm(C.byref(d), None) == 1
    assert b"null pointer" in lib.pf_last_error_string()
    a = _lib.AttnDesc()
    assert lib.pf_attention(C.byref(a), None) == 1


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of pf_conv_desc / pf_attn_desc are pinned to the C header: a probe compiled by gcc
    against include/panfusion_hip.h prints sizeof and offsetof of every field; they must equal ctypes'."""
    import subprocess
    structs = {"pf_conv_desc": _lib.ConvDesc, "pf_attn_desc": _lib.AttnDesc, "pf_attn_bwd_desc": _lib.AttnBwdDesc,
               "pf_linear_ws_desc": _lib.LinearWsDesc}
    lines = []
    for cname, cls in structs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "panfusion_hip.h"\nint main(void) {\n%s\nreturn 0; }\n'
                   % "\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True, timeout=120)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=60).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        cname, field, value = line.split()
        cls = structs[cname]
        want = C.sizeof(cls) if field == "sizeof" else getattr(cls, field).offset
        assert int(value) == want, "%s.%s: header %s, ctypes %d" % (cname, field, value, want)
        seen += 1
    # every ctypes field exists in the header (the probe would not compile otherwise) and nothing was skipped
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
    # and the header has no field the binding misses: the sizes agree, so a missing trailing field would show
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "panfusion_hip.h")).read(), flags=re.S)
    for cname, cls in structs.items():
        end = hdr.index("} %s;" % cname)
        body = hdr[hdr.rindex("typedef struct {", 0, end):end]
        names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:,|;)", body)
        assert sorted(names) == sorted(f for f, _ in cls._fields_), (cname, names)


def _conv_desc(**kw):
    """A structurally valid 1x1 problem on fake (never dereferenced) device addresses; kw overrides fields."""
    d = _lib.ConvDesc()
    d.a0, d.w, d.out = 0x10000, 0x20000, 0x30000
    d.c0, d.a0_ld = 64, 64
    d.n_img, d.h_in, d.w_in, d.h_out, d.w_out = 1, 1, 128, 1, 128
    d.ksize, d.stride, d.pad, d.upsample = 1, 1, 0, 0
    d.n_out, d.out_ld = 64, 64
    d.out_dtype = d.dtype = _lib.PF_BF16
    d.batch = 1
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_linear_ws_rejects_bad_arguments_before_any_launch():
    """pf_linear_ws validates shape, alignment and mode on the host (no GPU needed: every case fails before the launch)."""
    lib = _lib.lib()
    assert lib.pf_linear_ws_supported(163840, 960, 320, 3) == 1 and lib.pf_linear_ws_supported(163840, 640, 320, 3) == 0
    assert lib.pf_linear_ws_supported(163840, 2560, 320, 2) == 1 and lib.pf_linear_ws_supported(163840, 320, 640, 0) == 0
    assert lib.pf_linear_ws_supported(32, 320, 320, 0) == 0 and lib.pf_linear_ws_supported(4096, 300, 320, 0) == 0
    # the K = 640 / K = 1280 shapes: 256- (or 128-) channel workgroups, 16-bit and GEGLU outputs only
    S = lib.pf_linear_ws_supported
    assert S(40960, 5120, 640, 2) == 1 and S(40960, 1280, 640, 0) == 1 and S(40960, 640, 640, 0) == 1
    assert S(40960, 640, 640, 1) == 0 and S(40960, 640, 640, 2) == 0 and S(40960, 1920, 640, 3) == 0 and S(40960, 320, 640, 0) == 0
    assert S(10240, 10240, 1280, 2) == 1 and S(10240, 2560, 1280, 0) == 1 and S(16, 128, 1280, 0) == 1
    assert S(40960, 640, 640, 5) == 1 and S(10240, 1280, 1280, 5) == 1 and S(81920, 320, 320, 5) == 1 and S(40960, 320, 640, 5) == 0   # PF_LWS_VT
    assert S(10240, 1280, 1280, 1) == 0 and S(10240, 1280, 1280, 4) == 0 and S(10240, 320, 1280, 0) == 0 and S(8, 1280, 1280, 0) == 0

    def desc(**kw):
        d = _lib.LinearWsDesc()
        d.a, d.w, d.out, d.a_ld, d.out_ld = 0x10000, 0x20000, 0x30000, 320, 320
        d.M, d.N, d.K, d.dtype, d.mode = 4096, 320, 320, _lib.PF_F16, 0
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    for bad in (dict(K=640), dict(N=300), dict(a=0), dict(a=0x10002), dict(a_ld=324), dict(out_ld=300), dict(mode=7),
                dict(residual=0x40000), dict(mode=1, residual=0x40004, res_ld=320), dict(mode=3, N=960, out_ld=640),
                dict(mode=3, N=960, out_ld=640, out_vt=0x50000, rows_per_batch=100, vt_ld=4096), dict(dtype=_lib.PF_F32)):
        assert lib.pf_linear_ws(C.byref(desc(**bad)), None) != 0, bad
        assert lib.pf_last_error_string()


def test_conv_gemm_rejects_bad_arguments_before_any_launch():
    """Every rule of include/panfusion_hip.h's pf_conv_desc is checked on the host, before a launch (so this runs
    without a GPU): status PF_ERR_ARG and a message naming the problem."""
    lib = _lib.lib()
    bad = [
        (dict(c0=48), b"multiples of 64"),
        (dict(ksize=5), b"ksize"),
        (dict(stride=3), b"stride"),
        (dict(dtype=_lib.PF_F32), b"dtype"),
        (dict(out_dtype=_lib.PF_F16), b"out_dtype"),
        (dict(residual=0x40000, res_ld=64, res_dtype=_lib.PF_F16), b"res_dtype"),
        (dict(a0=0x10008), b"16-byte aligned"),
        (dict(out_ld=60), b"out_ld"),
        (dict(n_out=62, bias=0x50000), b"n_out"),
        (dict(h_out=2), b"output size"),
        (dict(batch=0), b"batch"),
        (dict(epilogue=7), b"epilogue"),
        (dict(epilogue=1, residual=0x40000, res_ld=64), b"GEGLU"),
        (dict(n_img=1, w_in=1 << 26, w_out=1 << 26, a0_ld=64), b"2 GiB"),
    ]
    for kw, needle in bad:
        assert lib.pf_conv_gemm(C.byref(_conv_desc(**kw)), None) == 1, kw
        assert needle in lib.pf_last_error_string(), (kw, lib.pf_last_error_string())
    # the split-K scratch query is pure host arithmetic: long K + few tiles wants scratch, a big grid does not
    assert lib.pf_conv_gemm_workspace_size(C.byref(_conv_desc(c0=5120, a0_ld=5120, w_in=256, w_out=256, n_out=1280, out_ld=1280))) > 0
    assert lib.pf_conv_gemm_workspace_size(C.byref(_conv_desc(w_in=163840, w_out=163840, n_out=320, out_ld=320))) == 0


def test_attention_and_norm_reject_bad_arguments():
    lib = _lib.lib()
    a = _lib.AttnDesc()
    a.q, a.k, a.vt, a.out = 0x10000, 0x20000, 0x30000, 0x40000
    a.dtype, a.B, a.H, a.D, a.nq, a.nk = _lib.PF_F16, 1, 1, 48, 64, 64
    a.q_ld = a.k_ld = a.o_ld = 64
    a.vt_ld = 64
    assert lib.pf_attention(C.byref(a), None) == 1 and b"head dim" in lib.pf_last_error_string()
    a.D, a.vt_ld = 64, 32
    assert lib.pf_attention(C.byref(a), None) == 1 and b"vt_ld" in lib.pf_last_error_string()
    a.vt_ld, a.bias = 64, 0x50000                                  # bias without flags
    assert lib.pf_attention(C.byref(a), None) == 1 and b"bias and flags" in lib.pf_last_error_string()
    a.flags, a.flags_ld, a.bias_ld, a.nk = 0x60000, 4, 64, 62        # bias rows are read as float4: nk must be a multiple of 4
    assert lib.pf_attention(C.byref(a), None) == 1 and b"nk % 4" in lib.pf_last_error_string()
    assert lib.pf_layernorm(0x10000, None, 0, _lib.PF_F16, 4, 4100, 0x20000, 0x30000, 1e-5, _lib.PF_F16, 0x40000, None) == 1
    assert lib.pf_layernorm(0x10000, None, 0, _lib.PF_F16, 4, 64, 0x20000, 0x30000, 1e-5, _lib.PF_BF16, 0x40000, None) == 1   # 16-bit in != out
    # scale without shift; fp32 output cannot be a split pair
    assert lib.pf_scale_shift_act(0x10000, 64, None, 0, _lib.PF_F32, 1, 16, 0x20000, None, 0, _lib.PF_F16, 0, 0x40000, None) == 1
    assert lib.pf_scale_shift_act(0x10000, 64, None, 0, _lib.PF_F32, 1, 16, None, None, 0, _lib.PF_F32, 1, 0x40000, None) == 1
    assert lib.pf_add(0x10000, 7, 0x20000, _lib.PF_F16, 16, 0x30000, None) == 1
    assert lib.pf_groupnorm_stats(0x10000, 60, None, 0, _lib.PF_F16, 1, 16, 32, 1e-5, 0x20000, 0x30000, 0x40000, 0x50000,
                                  0x60000, 1 << 20, None) == 1
    assert lib.pf_geglu(0x10000, _lib.PF_F16, 4, 12, 0x20000, None) == 1


def test_backward_entry_points_reject_bad_arguments():
    lib = _lib.lib()
    d = _lib.AttnBwdDesc()
    for i, f in enumerate(("q", "k", "v", "dout", "qt", "kt", "dot", "dq", "dk", "dv", "lse", "delta")):
        setattr(d, f, 0x10000 * (i + 1))
    d.dtype, d.B, d.H, d.D, d.nq, d.nk = _lib.PF_F16, 1, 2, 32, 64, 96
    d.q_ld = d.k_ld = d.v_ld = d.do_ld = d.dq_ld = d.dk_ld = d.dv_ld = 64
    d.qt_ld = d.dot_ld = 64
    d.kt_ld = 96
    d.scale = 32 ** -0.5
    d.kt_ld = 64                                                    # K^T must cover nk tokens
    assert lib.pf_attention_bwd(C.byref(d), None) == 1 and b"cover the token count" in lib.pf_last_error_string()
    d.kt_ld, d.D = 96, 48
    assert lib.pf_attention_bwd(C.byref(d), None) == 1 and b"head dim" in lib.pf_last_error_string()
    d.D, d.bias = 32, 0x70000                                       # bias without flags
    assert lib.pf_attention_bwd(C.byref(d), None) == 1 and b"bias and flags" in lib.pf_last_error_string()
    d.bias, d.lse = None, None
    assert lib.pf_attention_bwd(C.byref(d), None) == 1 and b"null pointer" in lib.pf_last_error_string()
    assert lib.pf_attention_delta(0x10000, 0x20000, _lib.PF_F16, 1, 2, 32, 64, 60, 64 * 60, 0x30000, None) == 1
    assert lib.pf_layernorm_bwd(0x10000, None, 0, _lib.PF_F32, 8, 4100, 0x20000, 1e-5, 0x30000, None, 0x40000, 0x50000, None) == 1
    assert lib.pf_layernorm_bwd(0x10000, None, 0, _lib.PF_U8, 8, 64, 0x20000, 1e-5, 0x30000, None, 0x40000, 0x50000, None) == 1
    assert lib.pf_layernorm_bwd_parts(10) == 3 and lib.pf_layernorm_bwd_parts(10 ** 7) == 512
    assert lib.pf_geglu_bwd(0x10000, 0x20000, _lib.PF_F16, 4, 12, 0x30000, None) == 1
    need = lib.pf_colsum_workspace_size(100000, 320)
    assert need == 64 * 320 * 4
    assert lib.pf_colsum(0x10000, _lib.PF_F16, 100000, 320, 320, 0x20000, 0x30000, need - 4, None) == 1
    assert b"workspace too small" in lib.pf_last_error_string()
    assert lib.pf_colsum(0x10000, _lib.PF_F16, 100000, 320, 300, 0x20000, 0x30000, need, None) == 1     # ld < N
    assert lib.pf_scale_f32(0x10000, 16, 0x20000, 4, _lib.PF_F32, 0x30000, None) == 1                  # state index
    assert lib.pf_pow2_scale(None, None) == 1


def test_pf_hip_lib_selects_another_build(tmp_path):
    """PF_HIP_LIB (same-box A/B of kernel builds, tools/gpu_lib_ab.sh) replaces the in-tree library path;
    a missing file must fail loudly -- there is no fallback."""
    import shutil
    import subprocess
    import sys
    other = tmp_path / "libpf_other.so"
    shutil.copy(_lib.LIB_PATH, other)
    code = "from panfusion_amd import _lib; print(_lib.LIB_PATH); print(_lib.lib().pf_version())"
    env = dict(os.environ, PF_HIP_LIB=str(other), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == str(other) and int(out.stdout.split()[1]) >= 100
    env["PF_HIP_LIB"] = str(tmp_path / "nope.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
