"""Diagnostic: where the time between the launches of the captured token step goes.  Reads a rocprofv3 --kernel-trace CSV of tests/diag/ar_ab.py and prints, averaged over
the steady-state tokens of the last generation: kernel time, gaps between consecutive launches inside a token, and the gap across the graph-launch boundary (last launch
of token n -> first launch of token n+1).
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -- python3 <repo>/tests/diag/ar_ab.py 1 ; python3 tests/diag/ar_gaps.py <out>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
	rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
	for k in ("k_sample_step", "k_gemv", "k_attn_decode", "k_layernorm", "copyBuffer", "k_gemm", "k_attn_fwd", "k_skinny"):
		if k in n: return k
	return n[:30]
# tokens are delimited by k_sample_step
idx = [i for i, r in enumerate(rows) if "k_sample_step" in r[2]]
idx = idx[-240:-10]                       # steady-state tokens of the last generation
tok_wall, tok_kern, gaps_in, gap_cross, per = [], [], [], [], collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
	seg = rows[a + 1:b + 1]               # launches of one token: first decode launch ... its sample step
	prev_end = rows[a][1]
	tok_wall.append(seg[-1][1] - rows[a][1])
	tok_kern.append(sum(e - s for s, e, _ in seg))
	for j, (s, e, n) in enumerate(seg):
		g = s - prev_end
		(gap_cross if j == 0 else gaps_in).append(g)
		per[short(n)].append((e - s, g))
		prev_end = e
n = len(tok_wall)
print(f"tokens {n}: wall per token {sum(tok_wall) / n / 1e3:.1f} us, kernel time {sum(tok_kern) / n / 1e3:.1f} us, launches per token {len(gaps_in) / n + 1:.0f}")
print(f"gap across the graph-launch boundary (sample step -> first launch of the next token): mean {sum(gap_cross) / n / 1e3:.2f} us, max {max(gap_cross) / 1e3:.2f}")
print(f"gaps inside a token: mean {sum(gaps_in) / len(gaps_in) / 1e3:.2f} us, total {sum(gaps_in) / n / 1e3:.1f} us per token")
for k, v in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
	print(f"  {k:16s} x{len(v) / n:6.1f} per token: duration mean {sum(d for d, _ in v) / len(v) / 1e3:6.2f} us, gap in front mean {sum(g for _, g in v) / len(v) / 1e3:5.2f} us")
