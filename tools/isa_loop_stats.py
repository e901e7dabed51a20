#!/usr/bin/env python
"""Instruction mix of the INNERMOST loops (>= --min instructions) of each kernel in a gfx950 .s file.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \
        -o /tmp/nt.s dynamont_amd/csrc/nt_kernels.hip
    python tools/isa_loop_stats.py /tmp/nt.s k_read_queueILi1 [--min 150]

The read-queue kernel holds the row loops of the backward and the forward sweep (and the short loops of
the traceback); each is reported with its size and mix, AGPR moves and SGPR spill traffic listed apart.
"""
import re
import sys
from collections import Counter


def loops_of(lines):
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    spans = []
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\S+)|s_branch (\S+)', l)
        if mm:
            tgt = mm.group(1) or mm.group(2)
            if tgt in labels and labels[tgt] < i:
                spans.append((labels[tgt], i))
    inner = [(a, b) for (a, b) in spans if not any((a <= c and d <= b) and (c, d) != (a, b) for (c, d) in spans)]
    return inner


def main(path, names, min_size):
    s = open(path).read()
    for name in names:
        m = re.search(r'^(_ZN4dynk\d+' + name + r'[^:\n]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
        if not m:
            print(name, 'not found')
            continue
        lines = [l.split(';')[0].strip() for l in m.group(2).split('\n')]
        lines = [l for l in lines if l and (l.endswith(':') or not l.startswith('.'))]
        for a, b in loops_of(lines):
            loop = [l for l in lines[a:b + 1] if not l.endswith(':')]
            if len(loop) < min_size:
                continue
            c = Counter()
            for l in loop:
                op = l.split()[0]
                if 'accvgpr' in op:
                    c['agpr:' + op] += 1
                elif op in ('v_writelane_b32', 'v_readlane_b32'):
                    c['lane:' + op] += 1
                elif 'f64' in op:
                    c['f64:' + op] += 1
                elif op.startswith('v_'):
                    c['v32:' + op] += 1
                elif op.startswith('s_'):
                    c['s:' + op] += 1
                elif op.startswith(('global', 'buffer', 'flat', 'scratch')):
                    c['mem:' + op] += 1
                elif op.startswith('ds_'):
                    c['lds:' + op] += 1
                else:
                    c['other:' + op] += 1
            tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
            valu = tot('f64') + tot('v32') + tot('agpr') + tot('lane')
            print(f"{name}: loop @{a} {len(loop)} instrs | VALU {valu} (f64 {tot('f64')} v32 {tot('v32')} agpr {tot('agpr')} "
                  f"lane {tot('lane')}) salu {tot('s:')} mem {tot('mem')} lds {tot('lds')}")
            for v, k in sorted(((v, k) for k, v in c.items()), reverse=True)[:18]:
                print(f"      {v:5d} {k}")


if __name__ == '__main__':
    args = sys.argv[1:]
    mn = 150
    if '--min' in args:
        i = args.index('--min')
        mn = int(args[i + 1])
        del args[i:i + 2]
    main(args[0], args[1:], mn)
