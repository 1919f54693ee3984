#!/usr/bin/env python
"""Static instruction budget of the hot loops (no GPU needed): compile one .hip of dinounet_amd/csrc to gfx950 assembly with the
library's own flags and, per kernel, count the instructions of its heaviest loop by issue class -- MFMA, VALU (transcendental / packed /
other), LDS, VMEM (global / buffer, LDS-DMA), scalar, waits and barriers.  The companion of tools/pmc_kernels.py: the counters say
where the cycles go, this says what the loop is made of (e.g. attention at d_head 64: 16 MFMA against ~140 VALU per 64-key tile,
32 of them quarter-rate v_exp_f32).

usage: python tools/isa_budget.py attention.hip [kernel-name-substring ...]      (writes nothing; prints a table)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dinounet_amd import _build  # noqa: E402

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("ds_", )):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop") or op.startswith("s_setprio") or op.startswith("s_sleep"):
        return "nop/prio"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(TRANS):
        return "valu.trans"
    if op.startswith("v_pk_"):
        return "valu.packed"
    if op.startswith("v_cvt"):
        return "valu.cvt"
    if op.startswith("v_"):
        return "valu.other"
    return "other"


def kernels(asm):
    """name -> list of lines of the function body"""
    out, cur = {}, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln)
    return out


def loops(body):
    """(start, end) line ranges of natural loops: a label and the last backward branch to it"""
    pos = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            pos[m.group(1)] = i
    res = {}
    for i, ln in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            s = pos[m.group(1)]
            res[s] = max(res.get(s, 0), i)
    return sorted(res.items())


def histogram(lines):
    h, ops = collections.Counter(), collections.Counter()
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        op = ln.split()[0]
        h[classify(op)] += 1
        ops[op] += 1
    return h, ops


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src = sys.argv[1]
    want = sys.argv[2:]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, *[f for f in _build.FLAGS if f != "-fPIC"], *_build.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S",
               os.path.join(_build.CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    order = ["mfma", "valu.trans", "valu.packed", "valu.cvt", "valu.other", "lds", "vmem", "salu", "waitcnt", "barrier", "nop/prio", "other"]
    print(f"# {src}: every loop that holds MFMAs (in program order: steady-state loop first, then peeled / tail loops), else the largest loop; "
          f"static counts per loop body, all paths of the body included")
    print("# " + " ".join(f"{c:>11}" for c in order) + "  kernel / loop")
    seen = set()
    for name, body in kernels(asm).items():
        pretty = re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", demangle(name))).split("(")[0]
        if want and not any(w in pretty for w in want):
            continue
        sig = hash("\n".join(body))
        if sig in seen:                                   # the same body listed under an alias symbol
            continue
        seen.add(sig)
        found = []
        for s_, e in loops(body):
            h, ops = histogram(body[s_:e + 1])
            found.append((s_, h, ops))
        if not found:
            continue
        with_mfma = [f for f in found if f[1]["mfma"]]
        show = with_mfma[:4] if with_mfma else [max(found, key=lambda f: sum(f[1].values()))]
        print(f"{pretty[:120]}")
        for s_, h, ops in show:
            print("  " + " ".join(f"{h[c]:11d}" for c in order) + f"  loop at line {s_}")
            top = ", ".join(f"{n} {o}" for o, n in ops.most_common(9) if not o.startswith("s_nop"))
            print(f"      top: {top}")


if __name__ == "__main__":
    main()
