#!/usr/bin/env python3
"""Cost of each phase of the cooperative step kernel, measured by REPEATING one phase.

Removing a phase changes what the later phases see; running it twice does not.  For each phase X this script
writes a copy of epi_gfx950.hip in which phase X runs 1 + EPI_REPS times (environment variable read by the
library; all but the last repetition discard their results) and builds it as csrc/variants/libepi_rep<X>.so.
Time the variants with
    EPI_GFX950_LIB=.../variants/libepi_repB.so EPI_REPS=1 python3 bench.py --steps 5 --warmup 17 --no-cpu
and subtract the EPI_REPS=0 figure of the same variant: the difference is the phase's cost.

The substitutions below are anchored on the kernel's text as it was when the per-phase figures of the design notes
were taken (before the record loads, the fused first attempt and the specialised builds); on a later text the
script stops at the first anchor it cannot find, which then needs bringing up to date.
"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "epidemicmodeling_amd", "csrc")
src = open(os.path.join(CSRC, "epi_gfx950.hip")).read()


def sub(s, old, new, count=1):
    assert s.count(old) >= 1, old
    return s.replace(old, new, count)


def common(s):
    s = sub(s, 'static int g_screen = 1;', 'static int g_screen = 1;\n#include <cstdlib>\nstatic int epi_reps() { static int r = -1; if (r < 0) { const char* e = getenv("EPI_REPS"); r = e ? atoi(e) : 0; } return r; }')
    s = s.replace('A.screen = g_screen;', 'A.screen = g_screen | (epi_reps() << 4);')
    # only the cooperative kernel decodes the repetition count
    head, tail = s.split('void k_step_coop(const StepArgs A, const DevParams* __restrict__ Dp)\n{', 1)
    tail = sub(tail, 'const bool screen = A.screen != 0;', 'const bool screen = (A.screen & 1) != 0;\n    const int reps = 1 + (A.screen >> 4);')
    head = head.replace('const bool screen = A.screen != 0;', 'const bool screen = (A.screen & 1) != 0;')
    return head + 'void k_step_coop(const StepArgs A, const DevParams* __restrict__ Dp)\n{' + tail


REP = 'for (int rep = 0; rep < reps; rep++) { const bool last = rep == reps - 1; (void)last;\n'


def phase_A(s):
    s = sub(s, '    if (active) {\n        S = in[src]; R = in[is + src]; R_pre = R;', '    ' + REP + '    suminf_pre = 0;\n    if (active) {\n        S = in[src]; R = in[is + src]; R_pre = R;')
    s = sub(s, '            code = order_code(lw);\n        }\n    }\n    __syncthreads();', '            code = order_code(lw);\n        }\n    }\n    }\n    __syncthreads();')
    return s


def phase_B(s):
    s = sub(s, '    unsigned long long fm0 = 0ULL, fm1 = 0ULL;                                     // bins 0..31 / 32..63\n#pragma unroll 1\n    for (int i = 0; i < H; i++) {\n        if (active) {',
            '    unsigned long long fm0 = 0ULL, fm1 = 0ULL;\n    ' + REP + '#pragma unroll 1\n    for (int i = 0; i < H; i++) {\n        if (active) {')
    s = sub(s, '            slots[i * TPB + tid] = nn;\n            if (i < 32) fm0 |=', '            if (last) slots[i * TPB + tid] = nn; else failbits = 0u;\n            if (i < 32) fm0 |=')
    s = sub(s, '    {\n        const int lane = tid & 63;\n        const unsigned int mine = (unsigned int)(__popcll(fm0) + __popcll(fm1));',
            '    }\n    {\n        const int lane = tid & 63;\n        const unsigned int mine = (unsigned int)(__popcll(fm0) + __popcll(fm1));')
    return s


def phase_C(s):
    s = sub(s, '#pragma unroll 1\n            for (unsigned int e = tid; e < cnt; e += TPB) {\n                const unsigned short entry = q[e];\n                const int t = entry & 255, i = (entry >> 8) & 63, j = entry >> 14;',
            REP + '#pragma unroll 1\n            for (unsigned int e = tid; e < cnt; e += TPB) {\n                const unsigned short entry = q[e];\n                const int t = entry & 255, i = (entry >> 8) & 63, j = entry >> 14;')
    s = sub(s, '                const bool fail = k < 0;\n                const bool pushed = q_push(fail, entry, qnext, &qn[cur ^ 1]);', '                const bool fail = k < 0 && last;\n                const bool pushed = q_push(fail, entry, qnext, &qn[cur ^ 1]);')
    s = sub(s, '                if (k >= 0) slot_words[(i * TPB + t) * 2 + j] = flip ? n - k : k;\n            }\n', '                if (k >= 0 && last) slot_words[(i * TPB + t) * 2 + j] = flip ? n - k : k;\n            }\n            }\n')
    return s


def phase_D(s):
    s = sub(s, '#pragma unroll 1\n    for (int it = 0; it < H; it++) {\n        const int i = it == 0 ? H - 1 : it - 1;\n        bool gfail = false;',
            'const int32_t R_A = R;\n    ' + REP + '    R = R_A; o_sum = 0; sum_inf = 0;\n#pragma unroll 1\n    for (int it = 0; it < H; it++) {\n        const int i = it == 0 ? H - 1 : it - 1;\n        bool gfail = false;')
    s = sub(s, 'if (gamma_is_zero(shape)) gslots[i * TPB + tid] = 0.0;', 'if (gamma_is_zero(shape)) { if (last) gslots[i * TPB + tid] = 0.0; }')
    s = sub(s, 'if (gamma_attempt(G, zf, wu, v)) gslots[i * TPB + tid] = gamma_finish(G, v, P.theta[it], g, (uint32_t)it);\n                else { slots[i * TPB + tid] = make_int2(Ii, 0); gfail = true; }',
            'if (gamma_attempt(G, zf, wu, v)) { const double gv = gamma_finish(G, v, P.theta[it], g, (uint32_t)it); if (last) gslots[i * TPB + tid] = gv; }\n                else if (last) { slots[i * TPB + tid] = make_int2(Ii, 0); gfail = true; }')
    s = sub(s, '    // ---- E: gamma retry rounds', '    }\n    // ---- E: gamma retry rounds')
    return s


def phase_E(s):
    s = sub(s, '#pragma unroll 1\n            for (unsigned int e = tid; e < cnt; e += TPB) {\n                const unsigned short entry = q[e];\n                const int t = entry & 255, it = entry >> 8;',
            REP + '#pragma unroll 1\n            for (unsigned int e = tid; e < cnt; e += TPB) {\n                const unsigned short entry = q[e];\n                const int t = entry & 255, it = entry >> 8;')
    s = sub(s, 'const bool pushed = q_push(!ok, entry, qnext, &qn[cur ^ 1]);\n                if (ok) gslots[i * TPB + t] = gamma_finish(G, v, P.theta[it], gt, (uint32_t)it);\n                else if (!pushed) gslots[i * TPB + t] = gamma_finish_from(G, P.theta[it], gt, (uint32_t)it, round + 1u);\n            }\n',
            'const bool pushed = q_push(!ok && last, entry, qnext, &qn[cur ^ 1]);\n                if (ok) { const double gv = gamma_finish(G, v, P.theta[it], gt, (uint32_t)it); if (last) gslots[i * TPB + t] = gv; }\n                else if (!pushed && last) gslots[i * TPB + t] = gamma_finish_from(G, P.theta[it], gt, (uint32_t)it, round + 1u);\n            }\n            }\n')
    return s


def phase_G(s):
    """only the gamma part of phase D (normals, constants, attempt 0, scaling)"""
    s = sub(s, '            const double shape = P.k[it] * (double)Ii;                             // init.jl:71-78\n',
            '            ' + REP + '            const double shape = P.k[it] * (double)Ii;                             // init.jl:71-78\n')
    s = sub(s, '                else { slots[i * TPB + tid] = make_int2(Ii, 0); gfail = true; }\n            }\n',
            '                else { slots[i * TPB + tid] = make_int2(Ii, 0); gfail = true; }\n            }\n            }\n')
    return s


def phase_P(s):
    """only the Philox block of phase B"""
    s = sub(s, '            rng_block(g, K_BINOM0, (uint32_t)i, 0, w);\n',
            '            for (int rep = 0; rep < reps; rep++) rng_block(g, K_BINOM0, (uint32_t)i + (uint32_t)rep * 0u, 0, w);\n')
    return s


PH = {"G": phase_G, "P": phase_P, "A": phase_A, "B": phase_B, "C": phase_C, "D": phase_D, "E": phase_E}
which = sys.argv[1:] or list(PH)
flags = "-O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math --offload-arch=gfx950".split()
procs = []
for x in which:
    s = PH[x](common(src))
    s = s.replace('#include "epi_step.hpp"', '#include "../epi_step.hpp"').replace('#include "epi_rng.hpp"', '#include "../epi_rng.hpp"').replace('#include "../../include/epi.h"', '#include "../../../include/epi.h"')
    path = os.path.join(CSRC, "variants", f"rep{x}.hip")
    open(path, "w").write(s)
    procs.append((x, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-Rpass-analysis=kernel-resource-usage", "-o",
                                       os.path.join(CSRC, "variants", f"libepi_rep{x}.so"), path],
                                      stderr=subprocess.PIPE, text=True)))
for x, p in procs:
    err = p.communicate()[1]
    lines = err.splitlines()
    info = ""
    for k, l in enumerate(lines):
        if "k_step_coopILb1" in l and "Function Name" in l:
            info = " ".join(t.split("remark: ")[-1].strip() for t in lines[k + 1:k + 8] if "VGPRs:" in t or "ScratchSize" in t)
    print(x, "rc", p.returncode, info)
    if p.returncode:
        print(err[-2000:])
