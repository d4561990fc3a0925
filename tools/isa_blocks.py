"""Basic-block instruction counts of one kernel in a hipcc -S listing: python tools/isa_blocks.py file.s kernel_substr"""
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + "E") if sys.argv[2] + "E" in s else s.index(sys.argv[2])
i = s.index(":\n", i)
k = s[i:]; k = k[:k.index(".Lfunc_end")]
cur = "entry"; cnt = {cur: [0, 0, 0, []]}; order = [cur]
for l in k.split("\n"):
    t = l.split(";")[0].strip()
    if not t or (t.startswith(".") and not t.endswith(":")): continue
    if t.endswith(":"):
        cur = t[:-1]; cnt[cur] = [0, 0, 0, []]; order.append(cur); continue
    op = t.split()[0]
    cnt[cur][0 if op.startswith("v_") else 1 if op.startswith("s_") else 2] += 1
    if "branch" in op: cnt[cur][3].append(t.split()[-1])
for b in order:
    v, sx, m, br = cnt[b]
    print("%-12s v%4d s%4d m%3d -> %s" % (b, v, sx, m, " ".join(br)))
