"""Kernel sequence of one steady-state frame from a rocprofv3 --kernel-trace CSV, with run-length compression, to spot
glue kernels (fills, copies, elementwise) on the frame's critical path.   python tools/exp_frame_seq.py <dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "at::native" in n:
        n = "torch:" + n.split("at::native::")[-1].split("<")[0][:40] + "|" + (n.split("at::native::")[1].split("<")[0][:30] if n.count("at::native::") > 1 else "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:70], r.get("Queue_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_lora_merge_layers") and not r[2].startswith("k_lora_merge_layers_bwd")]
# a frame = from one LoRA merge launch to the next with renders in between (one launch merges both nets' layers; builds that
# merged per net launched two - then every second mark opens a frame); the last complete one of the timed region is taken
seq = None
step = 2 if len(marks) >= 4 and not any(r[2].startswith("k_render_bwd") for r in rows[marks[-3]:marks[-2]]) else 1
cands = marks[::step]
frames = [rows[a:b] for a, b in zip(cands, cands[1:]) if any(r[2].startswith("k_render_bwd") for r in rows[a:b])]
if frames:                          # the frame of median span (the timed region's frames outnumber warm-up and accounting passes)
    frames.sort(key=lambda f: f[-1][1] - f[0][0])
    seq = frames[len(frames) // 2]
if seq is None:
    raise SystemExit("no frame found")
cover, end = 0, seq[0][0]
idle = []
for r in seq:                       # union of the kernel intervals (kernels of different streams overlap)
    if r[0] > end:
        idle.append(((r[0] - end) / 1e3, r[2]))
        end = r[0]
    if r[1] > end:
        cover += r[1] - end
        end = r[1]
print("frame span %.3f ms, %d kernels, kernel time summed %.3f ms, GPU busy (union) %.3f ms" %
      ((seq[-1][1] - seq[0][0]) / 1e6, len(seq), sum(r[1] - r[0] for r in seq) / 1e6, cover / 1e6))
print("largest idle gaps (us, before kernel):", sorted(idle, reverse=True)[:12])
out = []
for r in seq:
    d = (r[1] - r[0]) / 1e3
    if out and out[-1][0] == r[2]:
        out[-1][1] += 1; out[-1][2] += d
    else:
        out.append([r[2], 1, d])
glue = 0.0
for n, c, d in out:
    tag = ""
    if n.startswith(("torch:", "__amd_rocclr")):
        glue += d; tag = "   <-- glue"
    print(f"{c:4d} x {n:70s} {d:9.1f} us{tag}")
print("glue kernels: %.1f us" % glue)
