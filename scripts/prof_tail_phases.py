"""Wall time of the recognizer tail's launch chains in one bench step of a rocprofv3 kernel trace (rocpd sqlite): for each kernel
family the span first start -> last end, the sum of durations and the number of launches (LSTM: 64 lstm_step launches in two
chains of 32; decoder: dec_fc_att / dec_gru alternating).

    python scripts/prof_tail_phases.py /tmp/pr/r_results.db [step_index]
"""
import sqlite3
import sys

db = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = sqlite3.connect(db)
ends = [r[0] for r in c.execute("select end from kernels where name like 'postprocess_words_kernel%' order by end")]
a, b = ends[step - 1], ends[step]
rows = c.execute("select start, end, name from kernels where start >= ? and end <= ? order by start", (a, b)).fetchall()
print(f"step {step}: wall {(b - a) / 1e6:.2f} ms, {len(rows)} kernels")
fams = {"lstm": ("lstm_step_kernel",), "decoder": ("dec_fc_att_kernel", "dec_gru_kernel"), "gc_attention": ("gc_attention_kernel",),
        "nms": ("nms_select_kernel",), "postprocess": ("postprocess_words_kernel", "text_argmax_kernel"),
        "roi_align": ("roi_align_rotated_kernel",)}
for fam, keys in fams.items():
    sel = [(s, e) for s, e, n in rows if any(k in n for k in keys)]
    if not sel:
        continue
    span = (max(e for _, e in sel) - min(s for s, _ in sel)) / 1e3
    busy = sum(e - s for s, e in sel) / 1e3
    # chain-internal gaps: consecutive launches of the family
    sel.sort()
    gaps = [max(0, sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 1)]
    gaps.sort()
    med = gaps[len(gaps) // 2] if gaps else 0.0
    print(f"{fam:13s} launches {len(sel):4d}  span {span:8.1f} us  sum of durations {busy:8.1f} us  median gap to the next launch {med:5.1f} us")
