#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop from hipcc's assembly (tools/dev_kernel.sh leaves it in /tmp/geoac_dev).

    tools/isa_mix.py /tmp/geoac_dev/dev-hip-amdgcn-amd-amdhsa-gfx950.s k_rk4 [--dump]

The hot loop = the innermost natural loop (backward branch to a label) with the most FP64 instructions.  For the stratified RK4 kernels
that is the step loop's common path: the rolled two-stage loop sits inside it and is counted TWICE (stages 1 and 2), everything else once -
the figure is the number of instructions one lane issues per accepted RK4 step when no rare block is taken (blocks reached only through
s_cbranch_execz / __builtin_expect-cold branches are laid out after the loop or skipped over and are not on the fall-through path).

One wave per SIMD issues one instruction every >= 4 cycles whatever its kind (a wave64 FP64 instruction occupies the pipe 4 cycles), so
`issue floor` = instructions x 4 cycles / 2.4 GHz; bench.py reports the measured step time against it as roofline.issue.
"""
import collections
import json
import re
import sys


def parse(path, want):
    txt = open(path).read().split("\n")
    start = None
    for i, l in enumerate(txt):
        m = re.match(r"^(_Z\w+):", l)
        if m and want in m.group(1) and start is None:
            start, name = i, m.group(1)
    if start is None:
        sys.exit(f"no function matching {want}")
    end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm"))
    return name, txt[start:end + 1]


def classify(op):
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_accvgpr"):
        return "v_accvgpr"
    if op.startswith(("v_mov", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
        return "v_mov/cndmask"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if re.search(r"_f64|_f64_", op) or op.startswith(("v_rcp_f64", "v_rsq_f64")):
        return "fp64"
    if op.startswith("v_"):
        return "valu_other"
    return "other"


def trace(ins, labels, head, tail, skip_execz, back_trips=1):
    """walk the loop body from `head` to the back edge at `tail` along the fall-through path: s_cbranch_execnz (a rare block, laid out out of
    line) is never taken; a backward branch inside (the rolled stage loop) is taken ONCE (two trips); a forward s_cbranch_execz over an in-line
    block is taken or not by `skip_execz`; other forward uniform branches (vcc / scc) are not taken unless their target lies beyond an
    out-of-line block the walk would otherwise run into"""
    cnt = collections.Counter()
    i, trips, guard = head, collections.Counter(), 0
    path = []
    while guard < 20000 and i < len(ins):
        guard += 1
        s = ins[i]
        op = s.split()[0]
        cnt[classify(op)] += 1
        path.append(i)
        if i == tail:
            break
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels:
            t = labels[m.group(1)]
            if op == "s_branch":
                if t == head:
                    break
                if t <= i:                               # unconditional backward branch: the rolled stage loop's latch
                    trips[i] += 1
                    if trips[i] > back_trips:
                        i += 1
                        continue
                i = t
                continue
            if t <= i:                                   # backward conditional: the rolled stage loop
                trips[i] += 1
                if trips[i] <= back_trips:
                    i = t
                    continue
            elif op == "s_cbranch_execz" and skip_execz and t - i < 120:
                i = t
                continue
            elif op in ("s_cbranch_vccnz", "s_cbranch_vccz", "s_cbranch_scc1", "s_cbranch_scc0") and t > i:
                # the exit of the rolled loop after its second trip: taken when we have been here before
                trips[i] += 1
                if trips[i] >= back_trips + 1:
                    i = t
                    continue
        i += 1
    return cnt, path


def main():
    if "--trace" in sys.argv:
        path, want = sys.argv[1], sys.argv[2]
        name, lines = parse(path, want)
        ins, labels = [], {}
        for l in lines:
            s = l.split(";")[0].rstrip()
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                labels[m.group(1)] = len(ins)
                continue
            s = s.strip()
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            ins.append(s)
        # the step loop: the unconditional backward branch whose span holds the most FP64 instructions
        best = None
        for i, s in enumerate(ins):
            m = re.match(r"^s_branch\s+(\.LBB\d+_\d+)", s)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                h = labels[m.group(1)]
                n = sum(1 for q in ins[h:i] if classify(q.split()[0]) == "fp64")
                score = float(n) ** 3 / float(i - h + 1) ** 2              # (dense in FP64: the step loop, not the loops around it that also hold the leg ends)
                if best is None or score > best[2]:
                    best = (h, i, score)
        h, t, _ = best
        trips = 1
        for a in sys.argv:
            if a.startswith("--loop="):
                h, t = (int(x) for x in a[7:].split(","))
            if a.startswith("--trips="):
                trips = int(a[8:])                        # times the rolled stage loop's back edge is taken (1: two stages rolled; 2: three)
        out = {"kernel": name, "step_loop": [h, t], "rolled_loop_back_edges_taken": trips}
        for key, skip in (("in_line_blocks_run", False), ("in_line_blocks_skipped", True)):
            cnt, pth = trace(ins, labels, h, t, skip, trips)
            tot = sum(cnt.values())
            out[key] = {"instructions": tot, "mix": dict(cnt.most_common()), "issue_floor_us_at_4_cycles": round(tot * 4 / 2.4e3, 4),
                        "fp64_only_floor_us": round(cnt["fp64"] * 4 / 2.4e3, 4)}
        print(json.dumps(out, indent=1))
        return
    path, want = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    name, lines = parse(path, want)
    # instruction list with labels
    ins, labels = [], {}
    for l in lines:
        s = l.split(";")[0].rstrip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        s = s.strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        ins.append(s)
    # backward branches -> loops [head, tail]
    loops = []
    for i, s in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    fp = [1 if classify(s.split()[0]) == "fp64" else 0 for s in ins]
    pre = [0]
    for f in fp:
        pre.append(pre[-1] + f)
    nfp = lambda a, b: pre[b + 1] - pre[a]          # noqa: E731
    # the step loop: the loop with the most FP64 instructions that contains at most one nested loop
    best = None
    for a, b in loops:
        inner = [(c, d) for c, d in loops if a <= c and d <= b and (c, d) != (a, b)]
        if len(inner) <= 1 and (best is None or nfp(a, b) > nfp(*best[:2])):
            best = (a, b, inner)
    a, b, inner = best
    # forward skips inside the loop: `s_cbranch_execz L` over a rare block - instructions between the branch and L are NOT on the common path when the
    # block is cold.  hipcc places __builtin_expect-cold blocks out of line already; in-line exec-masked blocks are counted (they are issued).
    cnt = collections.Counter()
    body = []
    for i in range(a, b + 1):
        w = 2 if any(c <= i <= d for c, d in inner) else 1
        k = classify(ins[i].split()[0])
        cnt[k] += w
        body.append((w, k, ins[i]))
    total = sum(cnt.values())
    out = {"kernel": name, "loop_static_instructions": b - a + 1, "nested_loop_instructions": sum(d - c + 1 for c, d in inner),
           "per_step_instructions": total, "mix": dict(cnt.most_common()),
           "fp64_issue_floor_us": cnt["fp64"] * 4 / 2.4e3, "issue_floor_us": total * 4 / 2.4e3}
    print(json.dumps(out, indent=1))
    if dump:
        for w, k, s in body:
            print(f"{w} {k:14s} {s}")


if __name__ == "__main__":
    main()
