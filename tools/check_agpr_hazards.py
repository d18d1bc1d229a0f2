#!/usr/bin/env python3
"""Static look at the one hazard the compiler cannot see in this library: kernels that keep their accumulators in AGPRs issue their MFMAs
from asm statements (gemm_bf16x3_w128_kernel, mlp_bf16x3_kernel), which are opaque to the hazard recogniser — a v_accvgpr_read / _mov the
register allocator places too close behind the last MFMA reads a register the matrix pipe has not written yet.  The kernels carry their
own wait states (s_nop 15 x 2 in the asm statement of the last MFMA of a tail step / phase).  This script compiles a .hip file to gfx950
assembly and, per kernel, walks the text: for every v_accvgpr_read / _mov it counts the wait states (one per instruction, N + 1 per
s_nop N) since the last asm-statement MFMA that wrote the same AGPR, and reports the minimum.
Text order is not execution order across branches, so this is a lint, not a proof: below 11 (what the 8-pass 16x16x32 bf16 MFMA needs)
it exits non-zero, below 19 (the 16-pass requirement) it prints LOOK.  __graft_entry__.build() runs it on both files after the build.
(It flagged the one broken build this repository has had: a fused-MLP build in which the allocator read a15 ONE slot behind its MFMA
at the phase-2 -> phase-1 boundary; the step came out 23 % wrong and different on every run.)

    python3 tools/check_agpr_hazards.py multimodal_diffusion_amd/csrc/gemm_bf16x3.hip [name filter]"""
import re, subprocess, sys, tempfile, os

def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    if src.endswith(".s"):                       # an assembly listing made earlier
        lines = open(src).read().split("\n")
    else:
        # the compiler and the flags of the build that ships: `make -n -p` prints the Makefile's variables (HIPCC / ARCH / CXXFLAGS, with the
        # caller's environment and command-line overrides applied), so the assembly inspected here is the object that gets linked
        mk = subprocess.run(["make", "-n", "-p", "-C", os.path.dirname(os.path.abspath(src))], capture_output=True, text=True).stdout
        var = {m.group(1): m.group(2).strip() for m in re.finditer(r"^(HIPCC|ARCH|CXXFLAGS)\s*[:?]?=\s*(.*)$", mk, re.M)}
        hipcc = os.environ.get("HIPCC") or var.get("HIPCC") or "/opt/rocm/bin/hipcc"
        arch = os.environ.get("ARCH") or var.get("ARCH") or "gfx950"
        flags = (var.get("CXXFLAGS") or "-O3 -std=c++17 -fPIC --offload-arch=$(ARCH)").replace("$(ARCH)", arch).split()
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
            if r.returncode:
                print(f"FAIL {hipcc} could not compile {src} to assembly (rc {r.returncode}):\n{r.stderr[-3000:]}")
                return 2
            lines = open(out).read().split("\n")
    # clock = wait states issued so far in this kernel; wrote[r] = clock at the last asm-statement MFMA that wrote AGPR r
    name, clock, wrote, worst, in_asm, n_asm_mfma = None, 0, {}, {}, False, 0
    for ln, l in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, clock, wrote, in_asm = m.group(1), 0, {}, False
            continue
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or name is None or flt not in name:
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            m = re.match(r"v_mfma\S+\s+a\[(\d+):(\d+)\]", t)
            if m and in_asm:                      # (MFMAs the compiler emits itself are padded by its hazard recogniser)
                n_asm_mfma += 1
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    wrote[r] = clock
            clock += 1
            continue
        if op in ("v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
            r = int(re.search(r"\ba(\d+)\s*$", t).group(1))
            if r in wrote:
                gap = clock - wrote[r] - 1
                w = worst.get(name)
                if w is None or gap < w[0]:
                    worst[name] = (gap, ln, t)
        clock += int(t.split()[1]) + 1 if op == "s_nop" else 1
    rc = 0
    if n_asm_mfma == 0:       # a file handed to this lint is expected to hold asm-statement MFMAs: finding none means the lint looked at nothing
        print(f"FAIL no asm-statement MFMA with an AGPR destination found in {src} (filter '{flt}'): the lint has nothing to check")
        return 2
    for k, (w, ln, t) in sorted(worst.items()):
        try:
            kn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        except OSError:                          # no binutils on this host: the mangled name will do
            kn = k
        # the asm-statement MFMAs of this library are all v_mfma_f32_16x16x32_bf16: 8 passes, 11 wait states required (16 passes: 19)
        tag = "FAIL" if w < 11 else "LOOK" if w < 19 else "ok  "
        rc |= w < 11
        print(f"{tag} min wait states behind an asm-statement MFMA before a read of its AGPR: {w:3d}  (line {ln}: {t})  {kn[:90]}")
    return int(rc)


if __name__ == "__main__":
    sys.exit(main())
