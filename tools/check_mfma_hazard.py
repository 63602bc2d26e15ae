"""Build-time guard for the hipcc MFMA -> VALU hazard found in round 3 (attention.hip, A_step comment).

hipcc's hazard recogniser pads the wait states between a v_mfma* and the first VALU / memory read of its destination along the
LAYOUT (fall-through) path only.  When a short wave-uniform branch sits between the two, the TAKEN path skips the instructions the
padding was counted over and the consumer reads stale accumulator elements (intermittently: it depends on what else the SIMD issues).

This script reads gfx950 assembly text (hipcc -S --cuda-device-only) and, for every MFMA, follows every path that contains at least
one TAKEN branch: if an instruction that reads (or overwrites) a register of the MFMA's destination is reached in fewer wait states
than the MFMA needs, it reports the kernel, the MFMA, the branch and the consumer.  Fall-through-only paths are the compiler's
business and are not re-checked.  Wait states are counted conservatively: every instruction counts 1, `s_nop N` counts N + 1.

Required wait states (LLVM GCNHazardRecognizer, XDL write VGPR -> VALU / VMEM / LDS / export read or write, by passes: 2 passes 5,
4 passes 7, 8 passes 11, 16 passes 19, one more on gfx950).  hipcc's own fall-through padding in this tree confirms the first row
used here: `v_mfma_f32_16x16x32_bf16` (4 passes) is followed by `s_nop 7` = 8 wait states before a dependent VALU.

usage:  python tools/check_mfma_hazard.py file.s [file.s ...]      exit code 1 when a hazard is found
        python tools/check_mfma_hazard.py --self-test                a broken and a padded hand-written stream
        python tools/check_mfma_hazard.py --compile a.hip [b.hip]    hipcc -S the files with build.sh's flags, then check them
"""
import os
import re
import subprocess
import sys
import tempfile



def need_wait_states(op):
    if "16x16" in op:
        return 8           # 4 passes + 3 + 1 (gfx950): 16x16x32 f16 / bf16 / f8 and the legacy K = 16 forms
    if "32x32x16" in op or "32x32x64" in op:
        return 12          # 8 passes
    return 20              # anything else: the 16-pass bound


REG_RANGE = re.compile(r"\b([va])\[(\d+):(\d+)\]")
REG_ONE = re.compile(r"\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG_RANGE.finditer(text):
        for k in range(int(m.group(2)), int(m.group(3)) + 1):
            out.add((m.group(1), k))
    for m in REG_ONE.finditer(REG_RANGE.sub(" ", text)):
        out.add((m.group(1), int(m.group(2))))
    return out


class Ins:
    __slots__ = ("op", "args", "line", "text")

    def __init__(self, op, args, line, text):
        self.op, self.args, self.line, self.text = op, args, line, text


def parse_functions(path):
    """-> {function name: (instructions, {label: index})}"""
    funcs, cur, name = {}, None, None
    labels = {}
    for ln, raw in enumerate(open(path), 1):
        s = raw.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):$", s)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L") and not lab.startswith("BB") and not lab.startswith(".Ltmp"):
                if cur is not None:
                    funcs[name] = (cur, labels)
                cur, name, labels = [], lab, {}
            elif cur is not None:
                labels[lab] = len(cur)
            continue
        if cur is None or s.startswith("."):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        if not re.match(r"^[svdgbfe][a-z0-9_]*", op):
            continue
        cur.append(Ins(op, parts[1] if len(parts) > 1 else "", ln, s))
        if op == "s_endpgm":
            funcs[name] = (cur, labels)
            cur, name, labels = None, None, {}
    if cur is not None and name is not None:
        funcs[name] = (cur, labels)
    return funcs


def cost(ins):
    if ins.op == "s_nop":
        try:
            return int(ins.args.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def touches(ins, dst):
    """does the instruction read or write one of the registers in dst (vector side only)?"""
    if ins.op.startswith("s_") and not ins.op.startswith("s_nop"):
        return False
    return bool(regs_of(ins.args) & dst)


def check_function(name, instrs, labels, report):
    n = len(instrs)
    for m, ins in enumerate(instrs):
        if not ins.op.startswith("v_mfma") and not ins.op.startswith("v_smfma"):
            continue
        first = ins.args.split(",")[0]
        dst = regs_of(first)
        need = need_wait_states(ins.op)
        # depth-first over (index, waited, took_branch)
        stack = [(m + 1, 0, None)]
        seen = set()
        while stack:
            idx, waited, via = stack.pop()
            while idx < n and waited < need:
                key = (idx, via is not None)
                if key in seen and waited >= 0:
                    break
                cur = instrs[idx]
                if via is not None and touches(cur, dst):
                    # another MFMA accumulating into the same registers is the hardware's own dependency, not this hazard
                    if not (cur.op.startswith("v_mfma") or cur.op.startswith("v_smfma")):
                        report.append((name, ins, via, cur, waited, need))
                    break
                if via is None and touches(cur, dst) and not cur.op.startswith("v_mfma"):
                    break                      # consumed on the fall-through path: the compiler's padding applies
                if cur.op == "s_endpgm":
                    break
                if cur.op.startswith("s_cbranch") or cur.op == "s_branch":
                    tgt = cur.args.strip().split()[0] if cur.args.strip() else ""
                    if tgt in labels:
                        stack.append((labels[tgt], waited + 1, cur))
                    if cur.op == "s_branch":
                        break
                seen.add(key)
                waited += cost(cur)
                idx += 1
    return report


def check_file(path):
    rep = []
    for name, (instrs, labels) in parse_functions(path).items():
        check_function(name, instrs, labels, rep)
    return rep


def fmt(rep, path):
    out = []
    for name, mf, br, use, waited, need in rep:
        out.append(f"{path}: {name}\n    line {mf.line}: {mf.text}\n    line {br.line}: {br.text}   <- taken branch\n"
                   f"    line {use.line}: {use.text}   <- touches the MFMA destination after {waited} of {need} wait states")
    return "\n".join(out)


SELF_BROKEN = """
kern_broken:
	v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[12:15], 0
	s_cbranch_scc1 .LBB0_2
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
	v_add_f32 v20, v21, v22
.LBB0_2:
	v_mul_f32 v30, v5, v31
	s_endpgm
"""
SELF_PADDED = SELF_BROKEN.replace("kern_broken", "kern_padded").replace("\ts_cbranch_scc1 .LBB0_2", "\ts_nop 15\n\ts_cbranch_scc1 .LBB0_2")


def self_test():
    ok = True
    for text, expect in ((SELF_BROKEN, 1), (SELF_PADDED, 0)):
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            f.write(text)
        rep = check_file(f.name)
        os.unlink(f.name)
        got = 1 if rep else 0
        print(f"self-test {'broken' if expect else 'padded'} stream: {'flagged' if got else 'clean'} ({'ok' if got == expect else 'WRONG'})")
        ok = ok and got == expect
    return 0 if ok else 1


def compile_to_asm(src, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", '-DUVIT_SRC_HASH="chk"',
           "-S", "--cuda-device-only", "-o", out, src] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--self-test":
        sys.exit(self_test())
    files = []
    if args and args[0] == "--compile":
        extra = [a for a in args[1:] if a.startswith("-")]
        for src in [a for a in args[1:] if not a.startswith("-")]:
            files.append((compile_to_asm(src, extra), src, True))
    else:
        files = [(a, a, False) for a in args]
    bad = 0
    for path, shown, tmp in files:
        rep = check_file(path)
        if rep:
            bad += len(rep)
            print(fmt(rep, shown))
        else:
            print(f"{shown}: no MFMA -> VALU hazard across a taken branch")
        if tmp:
            os.unlink(path)
    sys.exit(1 if bad else 0)
