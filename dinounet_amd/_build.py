"""Build libdinounet_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdinounet_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "gemm_glds.hip", "gemm_p8.hip", "gemm_rk.hip", "gemm_skinny.hip", "conv_halo.hip", "conv_strip.hip", "attention.hip", "norm.hip", "msda.hip", "elementwise.hip", "loss.hip", "optim.hip", "augment.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc"]
# the measurement tools' build: environment knobs (DU_CONV_STRIP, DU_SKINNY_FUSE_KMAX, ...) compiled in; the release library has none (csrc/common.h)
if os.environ.get("DINOUNET_DEBUG_KNOBS") == "1":
    FLAGS.append("-DDU_DEBUG_KNOBS")
# per-source extras.  attention.hip: MFMA results straight into VGPRs (the softmax reads every S^T accumulator with VALU ops; in the
# accumulator half of the register file each one costs a v_accvgpr_read and a second register: 194 -> 166 registers, 2 -> 3 waves / SIMD)
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv_strip.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# kernels of these files must not use scratch: a register spill in a GEMM / attention / conv main loop or epilogue is a silent multi-ms
# regression (seen: an address helper inlined into every epilogue put 576 bytes per lane of the 256 x 256 kernel on the stack, +7 ms per
# step, all tests green).  The build reads hipcc's resource-usage remarks and fails on ScratchSize > 0 there.
NO_SCRATCH = ("gemm_bf16.hip", "gemm_glds.hip", "gemm_p8.hip", "gemm_rk.hip", "gemm_skinny.hip", "conv_halo.hip", "conv_strip.hip", "attention.hip")
REMARK = "-Rpass-analysis=kernel-resource-usage"
# timing-ablation and cycle-probe instantiations of the attention kernel (tools/attn_ablate.py) may spill: they never run in the product
import re as _re
_ABLATION = _re.compile(
    r"attn_fwd_kernelILi\d+ELb1EEE"                                            # round-3 kernel, ablation instantiation
    r"|attn_fwd_w64_kernelILi\d+ELi\d+ELb1E"                                   # round-4 kernel with the cycle probe
    r"|attn_fwd_w64_kernelILi\d+ELi\d+ELb\dELi\d+ELi[1-9]\d*ELi\d+EEE"         # ... with ablation bits (template <DH, NQB, PROBE, OCC, ABL, PRIO>)
    r"|attn_fwd_pipe_kernelILb1ELi\d+EEE|attn_fwd_pipe_kernelILb\dELi[1-9]\d*EEE")  # slot-pipelined experiment: probe / ablation builds


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/dinounet_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library; returns its path."""
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), *([REMARK] if s in NO_SCRATCH else []), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print("[dinounet_amd build]", " ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        if s in NO_SCRATCH:
            name, spills = None, []
            for ln in out.decode().splitlines():
                if "Function Name:" in ln:
                    name = ln.split("Function Name:")[1].split("[")[0].strip()
                elif "ScratchSize [bytes/lane]:" in ln and int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0]) > 0 and not _ABLATION.search(name or ""):
                    spills.append(f"{name} ({ln.split('ScratchSize [bytes/lane]:')[1].split('[')[0].strip()} B/lane)")
            if spills:
                raise RuntimeError(f"{s}: kernels spill to scratch: " + "; ".join(spills[:6]))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.check_call(cmd)
    # a kernel whose host stub the compiler dropped (seen once: a helper-lambda call inside an LDS-DMA builtin's arguments) links as an
    # undefined file-local symbol and only fails at dlopen on the GPU box: catch it here
    und = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True).stdout
    # ... and a function of ours declared with a stale signature in another source file (C++ mangling: the reference to the old signature
    # stays undefined; seen once: du_gemm_nt_p8 gained a parameter)
    bad = [ln.split()[-1] for ln in und.splitlines() if "_GLOBAL__N_" in ln or ("_Z" in ln and "du_" in ln)]
    if bad:
        os.remove(LIB)
        raise RuntimeError("undefined file-local symbols in libdinounet_hip.so (dropped kernel stubs?): " + ", ".join(bad[:4]))
    open(stamp, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
