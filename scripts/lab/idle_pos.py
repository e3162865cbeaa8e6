import sqlite3, sys
sys.path.insert(0, 'scripts')
from trace_gaps import short
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
ks = [(short(n), s, e) for n, s, e in rows]
# position bookkeeping: kernels since last embed_tok_norm (token start), tokens since last big gap (>2 ms)
since_tok = 0; tok = 0; out = []
prev_end = ks[0][2]
for i in range(1, len(ks)):
    name, s, e = ks[i]
    g = (s - prev_end) / 1e3
    if name.startswith("embed_tok_norm"):
        since_tok = 0; tok += 1
    else:
        since_tok += 1
    if g > 2000: tok = 0
    if 15 <= g <= 2000 and ("gemv" in name or "decode_attn" in name) :
        out.append((tok, since_tok, round(g, 1), ks[i-1][0], name))
    prev_end = max(prev_end, e)
print(len(out), "gaps")
for o in out[:150]: print(o)
