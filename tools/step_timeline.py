"""One step of a bench run as a text timeline, from `rocprofv3 --kernel-trace --output-format csv`:
start -> end, duration, stream and kernel of every dispatch between two consecutive `k_collect_emit` launches (a step of
distributed.sharded_step ends with the emit), taken near the end of the run.  Also the time during which exactly ONE kernel
was running (the serial parts of the step) and the longest such stretches.
    python tools/step_timeline.py <..._kernel_trace.csv> [which step from the end, default 2]"""
import csv
import sys


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), short(r["Kernel_Name"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[3] == "k_collect_emit"]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(ends) < back + 1:
        sys.exit("not enough steps in the trace")
    a, b = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[a:b]
    t0 = step[0][0]
    print(f"# one step (dispatches {a}..{b - 1} of {len(rows)}) of the traced command: start -> end (us from the step's first "
          f"dispatch), duration, stream, kernel")
    for s, e, st, n in step:
        print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  stream {st}  {n}")
    # coverage: how long exactly one / no kernel was running
    ev = sorted([(s, 1) for s, _, _, _ in step] + [(e, -1) for _, e, _, _ in step])
    depth, last, alone, idle, stretch, cur = 0, ev[0][0], 0, 0, [], None
    for t, d in ev:
        if depth == 1:
            alone += t - last
            if cur is None:
                cur = last
        elif depth == 0:
            idle += t - last
        if depth == 1 and (depth + d) != 1 and cur is not None:
            stretch.append((t - cur, cur - t0))
            cur = None
        depth += d
        last = t
    total = step[-1][1] - t0
    print(f"# step {total / 1e3:.1f} us; exactly one kernel on the device for {alone / 1e3:.1f} us, none for {idle / 1e3:.1f} us")
    for dur, at in sorted(stretch, reverse=True)[:6]:
        names = [n for s, e, _, n in step if s - t0 <= at + dur / 2 <= e - t0]
        print(f"#   alone for {dur / 1e3:7.1f} us from {at / 1e3:8.1f} us: {', '.join(names)}")


if __name__ == "__main__":
    main()
