"""Static hazard audit of the inline-asm MFMAs of attention_bwd1.hip (attn_bwd1p_kernel).

hipcc pads no hazard around an inline-asm statement (cdna_hip_programming.md, section 5.7 item 2), so the kernel's issue order has
to keep the distances itself.  This script compiles the file to ISA (hipcc -S, no GPU needed) and walks every basic block of the
pipelined kernel:
  (a) a vector / accumulator-move instruction that WRITES a register an inline-asm MFMA reads as an operand must be at least
      2 wait states ahead of it;
  (b) a register an inline-asm MFMA WRITES must not be read or written by a non-MFMA instruction within 12 wait states
      (8-pass XDL result), nor read as the A / B operand of another MFMA within 12 (an MFMA taking it whole as C is free).
Wait states are counted as a LOWER bound: one per instruction in between, N + 1 for s_nop N; the check does not follow
branches (a block boundary counts as 0: conservative), and an s_waitcnt / s_barrier counts as one.
Exit code 1 and a listing if anything is closer than that.  Also: (c) nothing but inline asm touches the owned accumulators
a[224:255], (d) nothing hipcc generates uses M0 (the loop's LDS-DMA statements leave their own value in it).  Used by tests/test_host_cpu.py."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "mca-paper_amd", "csrc", "attention_bwd1.hip")
KERNEL = "attn_bwd1p_kernel"


def compile_isa() -> str:
    out = os.path.join(tempfile.mkdtemp(prefix="audit_bwd1_"), "bwd1.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++20", "-fPIC", "-munsafe-fp-atomics",
           "-fno-slp-vectorize", f"-I{os.path.dirname(SRC)}", "-S", "--cuda-device-only", SRC, "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout)
    return open(out).read()


def regs_of(tok: str):
    """'v[10:13]' -> {('v', 10), ...}; 'a5' -> {('a', 5)}; anything else -> empty"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def parse(line: str):
    """-> (mnemonic, [operand tokens]) of an instruction line, or None"""
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"):
        return None
    parts = line.split(None, 1)
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    ops = [o.split()[0] for o in ops if o]          # drop modifiers such as 'offset:16'
    return parts[0], ops


WRITES_FIRST = ("v_", "ds_read", "ds_load", "global_load", "scratch_load", "buffer_load", "flat_load")


def audit(text: str):
    start = text.index(f"_Z17{KERNEL}")
    end = text.index(".amdhsa_kernel", start) if ".amdhsa_kernel" in text[start:] else len(text)
    lines = text[start:end].split("\n")
    problems, n_mfma, n_asm_mfma = [], 0, 0
    block = []          # [(line no, mnemonic, reads, writes, is_asm_mfma, wait states of this instruction)]
    in_asm = False
    for ln, raw in enumerate(lines):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        if re.match(r"^\.?L?BB\d+_\d+:", s) or s.startswith("s_cbranch") or s.startswith("s_branch") or s.startswith("s_endpgm"):
            block = []
            continue
        p = parse(raw)
        if p is None:
            continue
        mn, ops = p
        ws = int(ops[0]) + 1 if mn == "s_nop" and ops and ops[0].isdigit() else 1
        is_mfma = mn.startswith("v_mfma")
        reads, writes = set(), set()
        if is_mfma:
            n_mfma += 1
            writes = regs_of(ops[0]); a_b = regs_of(ops[1]) | regs_of(ops[2]); c = regs_of(ops[3]) if len(ops) > 3 else set()
            reads = a_b | c
        elif mn.startswith(WRITES_FIRST) and ops:
            writes = regs_of(ops[0])
            for o in ops[1:]:
                reads |= regs_of(o)
        else:          # stores, ds_write, everything else: operands are reads
            for o in ops:
                reads |= regs_of(o)
        if is_mfma and in_asm:
            n_asm_mfma += 1
            dist = 0
            for (pl, pm, pr, pw, pasm, pws) in reversed(block):          # (a)
                if dist >= 2:
                    break
                if not pm.startswith("v_mfma") and pm.startswith(("v_",)) and (pw & reads):
                    problems.append(f"line {ln}: {mn} reads {sorted(pw & reads)[:2]}.. written {dist} wait state(s) earlier by '{pm}' (line {pl})")
                dist += pws
        # (b): this instruction against earlier asm MFMA results
        dist = 0
        for (pl, pm, pr, pw, pasm, pws) in reversed(block):
            if dist >= 12:
                break
            if pasm and pw:
                if is_mfma:
                    whole_c = (len(ops) > 3 and regs_of(ops[3]) == pw)
                    bad = (pw & a_b) or ((pw & (reads | writes)) and not whole_c and regs_of(ops[0]) != pw)
                else:
                    bad = pw & (reads | writes)
                if bad:
                    problems.append(f"line {ln}: '{mn}' touches {sorted(bad)[:2]}.. {dist} wait state(s) after the inline-asm MFMA that wrote them (line {pl})")
            dist += pws
        block.append((ln, mn, reads, writes, is_mfma and in_asm, ws))
    return problems, n_mfma, n_asm_mfma


OWNED = {("a", i) for i in range(224, 256)}          # the two dQ accumulators (ACC_CLOB_0 / ACC_CLOB_1 of attention_bwd1.hip)


def audit_owned(text: str):
    """(c) the dQ accumulators a[224:255] are the landing registers of asynchronous loads; hipcc believes a register written when
    the statement that names it ends.  So NOTHING outside an inline-asm statement may read or write them, anywhere in the kernel."""
    start = text.index(f"_Z17{KERNEL}")
    end = text.index(".amdhsa_kernel", start) if ".amdhsa_kernel" in text[start:] else len(text)
    problems, in_asm, n_asm = [], False, 0
    for ln, raw in enumerate(text[start:end].split("\n")):
        s_ = raw.strip()
        if s_.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s_.startswith(";;#ASMEND"):
            in_asm = False; continue
        p_ = parse(raw)
        if p_ is None:
            continue
        regs = set()
        for o in p_[1]:
            regs |= regs_of(o)
        if regs & OWNED:
            if in_asm:
                n_asm += 1
            else:
                problems.append(f"line {ln}: '{p_[0]}' (not inline asm) touches {sorted(regs & OWNED)[:2]}..")
    return problems, n_asm


def audit_m0(text: str):
    """(d) the loop's LDS-DMA statements write M0 and do not restore it (hipcc treats M0 as reserved and rewrites it itself ahead of
    its own users): nothing hipcc generates in this kernel may READ M0 behind them - here: no compiler instruction names m0 at all."""
    start = text.index(f"_Z17{KERNEL}")
    end = text.index(".amdhsa_kernel", start) if ".amdhsa_kernel" in text[start:] else len(text)
    problems, in_asm = [], False
    for ln, raw in enumerate(text[start:end].split("\n")):
        s_ = raw.strip()
        if s_.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s_.startswith(";;#ASMEND"):
            in_asm = False; continue
        code = raw.split(";")[0]
        if not in_asm and re.search(r"\bm0\b", code):
            problems.append(f"line {ln}: compiler instruction uses m0: {code.strip()}")
    return problems


if __name__ == "__main__":
    txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else compile_isa()
    probs, n, na = audit(txt)
    print(f"{KERNEL}: {n} MFMAs ({na} inline asm), {len(probs)} hazard(s)")
    for p in probs[:40]:
        print("  " + p)
    probs2, nl = audit_owned(txt)
    print(f"{KERNEL}: {nl} inline-asm instructions on the owned accumulators a[224:255], {len(probs2)} compiler instruction(s) touching them")
    for p in probs2[:20]:
        print("  " + p)
    probs3 = audit_m0(txt)
    print(f"{KERNEL}: {len(probs3)} compiler instruction(s) using m0")
    for p in probs3[:10]:
        print("  " + p)
    sys.exit(1 if (probs or probs2 or probs3) else 0)
