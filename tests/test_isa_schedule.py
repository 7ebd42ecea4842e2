"""The GEMM loops order their LDS reads, DMA, counted waits, barriers and MFMAs by hand; nothing in the
language stops the compiler from moving register-only MFMAs or LDS reads across a bare s_barrier.  These
tests compile the kernels to gfx950 assembly (no GPU needed) and check the instruction order of the
steady-state loop and of the two peeled K-tiles behind it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "omni-video_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
KINDS = ("v_mfma", "s_barrier", "ds_read", "s_waitcnt", "global_load_lds", "scratch_")


def listing(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


def kernels(text, name):
    """{mangled name: body} for every instantiation of `name`."""
    found = {}
    for m in re.finditer(r"^(_ZN\S*%s\S*):\s" % name, text, re.M):
        end = text.index(".end_amdhsa_kernel", m.end()) if ".end_amdhsa_kernel" in text[m.end():] else len(text)
        body = text[m.end():end]
        found[m.group(1)] = body[:body.index("s_endpgm")] if "s_endpgm" in body else body
    return found


def runs(lines):
    """Run-length sequence of the instruction kinds that matter, e.g. ['12 ds_read', '2 global_load_lds', ...]."""
    seq = []
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        op = ln.split()[0]
        kind = next((k for k in KINDS if op.startswith(k)), None)
        if kind is None:
            continue
        if seq and seq[-1][1] == kind:
            seq[-1][0] += 1
        else:
            seq.append([1, kind])
    return [f"{n} {k}" for n, k in seq]


def loop_and_tail(body):
    lines = body.split("\n")
    head = next(i for i, l in enumerate(lines) if "Inner Loop Header" in l)
    back = next(i for i in range(head, len(lines)) if "s_cbranch_scc1" in lines[i])
    # the two peeled K-tiles: from the loop's back branch to the last MFMA and the barriers that close it
    last = max(i for i in range(back, len(lines)) if lines[i].split(";")[0].strip().startswith("v_mfma"))
    end = last + 1
    while end < len(lines):
        op = lines[end].split(";")[0].strip().split()
        if op and any(op[0].startswith(k) for k in KINDS) and not op[0].startswith("s_barrier"):
            break
        end += 1
    return runs(lines[head:back + 1]), runs(lines[back + 1:end])


def phase(reads, dma, wait, mfma):
    out = []
    if reads:
        out.append(f"{reads} ds_read")
    if dma:
        out.append("2 global_load_lds")
    if wait:
        out.append("1 s_waitcnt")          # counted vmcnt in front of the barrier
    out.append("1 s_barrier")
    if reads:
        out.append("1 s_waitcnt")          # lgkmcnt behind it
    out += [f"{mfma} v_mfma", "1 s_barrier"]
    return out


def merge(seq):
    """Adjacent equal kinds of consecutive phases (two s_barrier in a row) merge in the run-length form."""
    out = []
    for item in seq:
        n, k = item.split(" ", 1)
        if out and out[-1][1] == k:
            out[-1][0] += int(n)
        else:
            out.append([int(n), k])
    return [f"{n} {k}" for n, k in out]


def expected(mfma):
    steady = phase(12, True, True, mfma) + phase(4, True, True, mfma) + phase(8, True, False, mfma) + \
        phase(0, True, True, mfma)
    penult = phase(12, True, True, mfma) + phase(4, True, True, mfma) + phase(8, False, False, mfma) + \
        phase(0, False, True, mfma)
    last = phase(12, False, True, mfma) + phase(4, False, True, mfma) + phase(8, False, False, mfma) + \
        phase(0, False, False, mfma)
    # one more barrier closes the loop for the wave group that started first
    return merge(steady), merge(penult + last + ["1 s_barrier"])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("src,kernel,mfma,count", [("gemm256.hip", "gemm256_kernel", 16, 4),
                                                   ("fp8.hip", "gemm_fp8p_kernel", 4, 5)])
def test_phased_gemm_loop_keeps_its_order(tmp_path, src, kernel, mfma, count):
    ks = kernels(listing(src, tmp_path), kernel)
    assert len(ks) == count, sorted(ks)
    want_loop, want_tail = expected(mfma)
    for name, body in ks.items():
        assert "scratch_" not in body, f"{name}: register spills"
        loop, tail = loop_and_tail(body)
        assert loop == want_loop, (name, loop)
        assert tail == want_tail, (name, tail)


def registers(token_text):
    """VGPR numbers named in an operand string: v12, v[8:11]."""
    regs = set()
    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", token_text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(n) for n in re.findall(r"\bv(\d+)\b", token_text))
    return regs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_gemm_bias_loads_hidden_from_the_compiler_are_not_touched_before_their_wait(tmp_path):
    """The bias values of the 256 x 256 GEMM are fetched by inline assembly inside the last K-tile (the compiler
    keeps no books on them) and waited for by an `s_waitcnt vmcnt(0)` in front of the epilogue.  Nothing the compiler
    emits in between may read or write the destination registers: a copy made before the data has landed would
    carry garbage and no test of values is guaranteed to see it."""
    ks = kernels(listing("gemm256.hip", tmp_path), "gemm256_kernel")
    assert len(ks) == 4, sorted(ks)
    for name, body in ks.items():
        lines = [ln.split(";")[0].strip() for ln in body.split("\n")]
        loads = [i for i, ln in enumerate(lines) if ln.startswith("global_load_dwordx2")]
        assert len(loads) == 4, (name, len(loads))
        dest = set()
        for i in loads:
            dest |= registers(lines[i].split(",")[0])
        assert len(dest) == 8, (name, sorted(dest))
        wait = next(i for i in range(loads[-1], len(lines)) if lines[i] == "s_waitcnt vmcnt(0)")
        for ln in lines[loads[-1] + 1:wait]:
            if not ln or ln.endswith(":") or ln.startswith((".", "s_")):
                continue
            assert not (registers(ln) & dest), (name, ln)


def main_loop(body):
    """Lines of the loop (label ... backward branch to it) that holds the most MFMAs; the shortest such loop."""
    lines = body.split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    mfma = [0]
    for ln in lines:
        mfma.append(mfma[-1] + ("v_mfma" in ln))
    best, key = (0, 0), (-1, 0)
    for i, ln in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and labels.get(m.group(1), i) < i:
            lo = labels[m.group(1)]
            k = (mfma[i + 1] - mfma[lo], lo - i)
            if k > key:
                best, key = (lo, i), k
    return lines[best[0]:best[1] + 1]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
@pytest.mark.parametrize("src,kernel,count", [("attn.hip", "attn_pipe_kernel", 2), ("attn.hip", "attn_i8_kernel", 2),
                                              ("vae.hip", "conv3d_cl_kernel", 8)])
def test_register_bound_loops_do_not_spill(tmp_path, src, kernel, count):
    """The attention loop sits at the 256-register limit and the convolution at 168 (three waves per SIMD): a
    change that tips either into scratch memory inside its loop costs far more than it can gain.  (Values
    that are only needed after the loop may be parked in scratch across it; that is once per workgroup.)"""
    text = listing(src, tmp_path)
    ks = kernels(text, kernel)
    assert len(ks) == count, sorted(ks)
    for name, body in ks.items():
        loop = main_loop(body)
        assert sum("v_mfma" in ln for ln in loop) >= 4, (name, len(loop))
        assert not any("scratch_" in ln for ln in loop), f"{name}: register spills inside the main loop"
    if kernel == "attn_pipe_kernel":
        w8 = next(b for n, b in ks.items() if "_w8" in n)
        assert "scratch_" not in w8, "the self-attention kernel spills"
    if kernel == "conv3d_cl_kernel":
        # the split-bf16 kernel with 96 output channels per workgroup is the one the large layers run
        meta = text[text.index("amdhsa.kernels"):]
        m = re.search(r"\.name:\s+\S*conv3d_cl_kernelILi3ELb1E\S*.*?\.vgpr_count:\s+(\d+)", meta, re.S)
        assert m and int(m.group(1)) <= 168, m and m.group(1)
