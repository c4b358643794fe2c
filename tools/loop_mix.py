"""Instruction mix of one kernel's main loop (smallest backward-branch span holding the most MFMAs) from an llvm-objdump -d listing.
   usage: llvm-objdump -d build/dev_attention.o | python tools/loop_mix.py <kernel-substring> [seq]"""
import re, sys, collections
pat = sys.argv[1]; seq = len(sys.argv) > 2
lines = sys.stdin.read().split('\n')
start = [i for i, l in enumerate(lines) if pat in l and l.endswith('>:')][0]
end = next((i for i in range(start + 1, len(lines)) if lines[i].endswith('>:')), len(lines))
ins = []
for l in lines[start:end]:
    m = re.match(r'\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):', l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
spans = []
for a, op, args in ins:
    if op.startswith('s_cbranch') or op == 's_branch':
        off = int(args.split()[-1])
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            spans.append((sum(1 for x in ins if tgt <= x[0] <= a and x[1].startswith('v_mfma')), -(a - tgt), tgt, a))
best = max(spans)[2:]          # the smallest backward-branch span holding the most MFMAs
loop = [x for x in ins if best[0] <= x[0] <= best[1]]
print(f"loop {best[0]:#x}..{best[1]:#x}: {len(loop)} instructions")
c = collections.Counter(op for _, op, _ in loop)
valu = sum(n for op, n in c.items() if op.startswith('v_') and not op.startswith('v_mfma'))
print("VALU", valu, "MFMA", sum(n for op, n in c.items() if op.startswith('v_mfma')), "waitcnt", c.get('s_waitcnt', 0))
for op, n in c.most_common(16): print(f"  {n:4d} {op}")
if seq:
    for a, op, args in loop: print(hex(a), op, args)
