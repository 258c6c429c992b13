"""For every MFMA loop of every kernel in an ISA listing (hipcc -S --cuda-device-only): its length, MFMA count, branches and the
histogram of the s_waitcnt vmcnt(N) values inside it.  A loop that keeps a ring of loads in flight shows only large N; a run
N, N-1, ... 0 is a ring drained once per iteration (a load inside a run-time branch, or a load whose value is needed at once).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only worddiffusion_amd/csrc/wd_ff.hip -o /tmp/wd_ff.s
    python tools/isa_loop_waits.py /tmp/wd_ff.s
"""
import re
import sys

for f in sys.argv[1:]:
    lines = open(f).read().split("\n")
    kstart = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*(;.*)?$", l)]
    kstart.append(len(lines))
    for a, b in zip(kstart, kstart[1:]):
        name = lines[a].split(":")[0]
        body = lines[a:b]
        for h, l in enumerate(body):
            if "Loop Header" not in l:
                continue
            k = h
            while k >= 0 and not re.match(r"^\.LBB\d+_\d+:", body[k]):
                k -= 1
            if k < 0:
                continue
            lab = body[k].split(":")[0]
            tag = "Header=" + lab[2:] + " "   # the blocks of the loop carry "in Loop: Header=BBn_m Depth=d"
            last = k
            for i in range(k + 1, len(body)):
                if re.match(r"^\.LBB\d+_\d+:|^; %bb\.", body[i]) and tag in body[i]:
                    last = i
            end = last + 1
            while end < len(body) and not re.match(r"^\.LBB\d+_\d+:|^; %bb\.|^\.Lfunc_end", body[end]):
                end += 1
            seg = body[k:end]
            nm = sum("v_mfma" in x for x in seg)
            if nm < 16:
                continue
            w = {}
            for x in seg:
                mm = re.search(r"vmcnt\((\d+)\)", x)
                if mm:
                    w[int(mm.group(1))] = w.get(int(mm.group(1)), 0) + 1
            print(f.split("/")[-1], name[:90], "| loop", lab, "lines", len(seg), "mfma", nm, "branches",
                  sum("s_cbranch" in x for x in seg), "vmcnt", dict(sorted(w.items())))
