"""Build check of the fp32 instantiation (libpigeon_hip_f32.so): no fp64 ARITHMETIC in the device code outside the kernels that own absolute time.

  python tools/check_f32_purity.py [pigeon.jl_amd/csrc/pg_api_f32.s]      (the .s comes from `make -C pigeon.jl_amd/csrc pg_api_f32.s`)

The kernels are written against the scalar type `real`; a stray `double` (or an unwrapped floating literal) would silently promote part of an
expression to fp64 in the fp32 build.  This reads the gfx950 assembly of that build and fails when a kernel issues an fp64 arithmetic
instruction it is not entitled to:
  * every kernel may convert / compare / frexp fp64 values (v_cvt_*, v_cmp_*, v_frexp_*: reading the double time grid, libm's sincosf reduction);
  * the kernels that compute with ABSOLUTE TIME (tdouble: k_time_steps and the k_project<true> instantiation that carries the time grid in the fused launch, the
    warm-branch interpolation weights of k_nodes / k_nodes_warm, t += dt of k_advance) may use fp64 arithmetic;
  * everything else (k_solve, k_linearize, k_limits, k_qp_dec, k_project, k_hji_*) must be pure fp32.
Exit status 0 = clean; 1 = violations (listed)."""
import collections
import os
import re
import sys

TIME_KERNELS = ("k_time_steps", "k_nodes", "k_nodes_linearize", "k_nodes_warm", "k_nodes_dec", "k_advance")
HARMLESS = re.compile(r"^v_(cvt_|cmp_|cmpx_|frexp_)")


def scan(path):
    per = collections.defaultdict(collections.Counter)
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        if cur and line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur:
            t = line.split()
            if t and re.match(r"^v_\w*f64", t[0]):
                per[cur][t[0]] += 1
    return per


def kernel_name(mangled):
    m = re.match(r"^_ZN2pg(\d+)", mangled)
    if not m:
        return mangled
    n = int(m.group(1)); s = mangled[len(m.group(0)):]
    return s[:n]


def violations(path):
    bad = {}
    for k, cnt in scan(path).items():
        name = kernel_name(k)
        if name in TIME_KERNELS or "k_projectILb1E" in k:          # k_project<true> hosts the time grid of the fused launch
            continue
        arith = {op: n for op, n in cnt.items() if not HARMLESS.match(op)}
        if arith:
            bad[name] = arith
    return bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "pigeon.jl_amd", "csrc", "pg_api_f32.s")
    bad = violations(p)
    for k, v in bad.items():
        print(f"fp64 arithmetic in the fp32 build of {k}: {v}")
    print("fp32 purity:", "FAILED" if bad else "ok")
    sys.exit(1 if bad else 0)
