#!/usr/bin/env python3
"""Build aid (no GPU): static instruction counts per basic block of one kernel, from the device assembly
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -ffp-contract=on -I include -I space_gym_amd/csrc -S --cuda-device-only -o sg.s space_gym_amd/csrc/sg_engine.hip
    python tools/asm_blocks.py sg.s 'goal_step_kernelILi3ELb0E'
"""
import re
import sys


def main():
    s = open(sys.argv[1]).read().splitlines()
    pat = re.compile(r'^(_Z\w*' + sys.argv[2] + r'\w*):')
    start = [i for i, l in enumerate(s) if pat.match(l)][0]
    end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
    blocks, cur = [], ['entry', []]
    for l in s[start + 1:end]:
        l = l.split(';')[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur); cur = [m.group(1), []]
            continue
        t = l.strip()
        if t.startswith('.'):
            continue
        cur[1].append(t)
        if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm'):  # a block ends at a branch as well
            blocks.append(cur); cur = ['  (fallthrough)', []]
    blocks.append(cur)
    tot = 0
    for name, ins in blocks:
        n = len(ins); tot += n
        v = sum(1 for i in ins if i.startswith('v_')); sa = sum(1 for i in ins if i.startswith('s_'))
        ds = sum(1 for i in ins if i.startswith('ds_'))
        g = sum(1 for i in ins if re.match(r'(global|buffer|scratch|flat)_', i))
        f64 = sum(1 for i in ins if re.match(r'v_\w+_f64', i))
        tr = sum(1 for i in ins if re.match(r'v_(rsq|rcp|sqrt|log|exp|sin|cos)_', i))
        br = [i for i in ins if i.startswith('s_cbranch') or i.startswith('s_branch')]
        print("%-12s n=%4d valu=%4d (f64 %3d trans %2d) salu=%4d lds=%3d vmem=%3d  %s" % (
            name, n, v, f64, tr, sa, ds, g, ' '.join(b.split()[0][2:] + '->' + b.split()[-1] for b in br)))
    print('total', tot)


if __name__ == "__main__":
    main()
