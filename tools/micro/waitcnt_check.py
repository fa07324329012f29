"""Static check of s_waitcnt placement in a gfx9-family kernel's assembly (hipcc -S output): every register read (or overwritten)
while a memory instruction that writes it may still be in flight is reported.  Model: VMEM loads and stores retire in order
(vmcnt), LDS instructions retire in order (lgkmcnt); scalar memory loads share lgkmcnt and may return out of order, so only
lgkmcnt(0) covers them.  The instruction stream is walked linearly (branches are not followed: code behind a forward branch is
checked as if the branch fell through, and the set of pending registers is dropped at every label and branch), which is exact
for the straight-line phases the hand-scheduled kernels consist of and says nothing about hazards that cross a branch.
usage: waitcnt_check.py file.s [kernel-name-substring ...]"""
import re
import sys

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    if re.search(r"\bvcc\b", tok): out += [("s", 106), ("s", 107)]
    return out


def check(body, name, verbose=True):
    vm_issued = lg_issued = 0          # instructions issued so far per counter
    vm_done = lg_done = 0              # retired for certain (from the last waits)
    pend = {}                          # register -> (counter, seq, kind, line)
    smem_out = []                      # seqs of outstanding scalar loads
    problems = []
    for ln, raw in enumerate(body):
        t = raw.split(";")[0].strip()
        if t.endswith(":") or t.startswith(("s_cbranch", "s_branch")):      # join / branch: what is pending beyond here depends on the path
            pend.clear()
            smem_out = []
            continue
        if not t or t.startswith("."): continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m: vm_done = max(vm_done, vm_issued - int(m.group(1)))
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                if smem_out and n != 0: pass                      # out-of-order returns: nothing certain
                else: lg_done = max(lg_done, lg_issued - n)
                if n == 0: smem_out = []
            continue
        is_vm_ld = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
        is_vm_st = op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic"))
        is_ds_rd = op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle")
        is_ds_wr = op.startswith("ds_write") or op.startswith("ds_add") or op.startswith("ds_max") or op.startswith("ds_min")
        is_smem = op.startswith("s_load") or op.startswith("s_buffer_load")
        lds_dma = is_vm_ld and ("_lds" in op or re.search(r"\blds\b", rest) is not None)
        has_dst = not (is_vm_st or is_ds_wr or op.startswith(("s_barrier", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_setprio", "s_sleep", "s_cmp", "s_bitcmp")) or lds_dma)
        if op.startswith("v_cmp") and not op.endswith("_e64"): has_dst = False      # writes vcc implicitly
        dst = regs(ops[0]) if (has_dst and ops) else []
        srcs = [r for o in (ops[1:] if has_dst else ops) for r in regs(o)]
        if op.startswith("v_cmp") and not op.endswith("_e64"): dst = [("s", 106), ("s", 107)]
        # accumulating forms read their destination
        if op.startswith(("v_fmac", "v_mac", "v_dot")) or (op.startswith("v_mfma") and False): srcs += dst

        def ready(r):
            p = pend.get(r)
            if not p: return True
            ctr, seq = p[0], p[1]
            ok = (seq <= vm_done) if ctr == "vm" else (seq <= lg_done and (p[2] != "smem" or seq not in smem_out))
            if ok: del pend[r]
            return ok
        for r in srcs:
            if not ready(r): problems.append((ln, "read of %s%d before its %s (line %d) is waited for" % (r[0], r[1], pend[r][2], pend[r][3]), t))
        for r in dst:
            if not ready(r): problems.append((ln, "overwrite of %s%d while its %s (line %d) is in flight" % (r[0], r[1], pend[r][2], pend[r][3]), t))
        if is_vm_ld or is_vm_st:
            vm_issued += 1
            if is_vm_ld and not lds_dma:
                for r in dst: pend[r] = ("vm", vm_issued, op, ln)
        elif is_ds_rd or is_ds_wr:
            lg_issued += 1
            if is_ds_rd:
                for r in dst: pend[r] = ("lg", lg_issued, op, ln)
        elif is_smem:
            lg_issued += 1
            smem_out.append(lg_issued)
            for r in dst: pend[r] = ("lg", lg_issued, "smem", ln)
    if verbose:
        print("%s: %d instructions, %d VMEM, %d LDS/SMEM, %d finding(s)" % (name[:90], len(body), vm_issued, lg_issued, len(problems)))
        for p in problems[:12]: print("   line %d: %s   [%s]" % p)
    return problems


def kernels(asm):
    for m in re.finditer(r"^(_Z\S+):[^\n]*$", asm, flags=re.M):
        body = asm[m.end():]
        e = body.find("s_endpgm")
        if e < 0: continue
        yield m.group(1), body[:e].split("\n")


if __name__ == "__main__":
    asm = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    total = 0
    for nm, body in kernels(asm):
        if pats and not any(p in nm for p in pats): continue
        total += len(check(body, nm))
    sys.exit(1 if total else 0)
