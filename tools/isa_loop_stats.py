#!/usr/bin/env python
"""Count the instruction mix of the largest loop of each kernel in a gfx950 .s file.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \
        -o /tmp/nt.s dynamont_amd/csrc/nt_kernels.hip
    python tools/isa_loop_stats.py /tmp/nt.s k_backwardILb1 k_forwardILb1
"""
import re
import sys
from collections import Counter


def main(path, names):
    s = open(path).read()
    for name in names:
        m = re.search(r'^(_ZN4dynk\d+' + name + r'[^:\n]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M)
        if not m:
            print(name, 'not found')
            continue
        lines = [l.split(';')[0].strip() for l in m.group(2).split('\n')]
        lines = [l for l in lines if l and (l.endswith(':') or not l.startswith('.'))]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
        best = None
        for i, l in enumerate(lines):
            mm = re.match(r's_cbranch_\w+ (\S+)|s_branch (\S+)', l)
            if mm:
                tgt = mm.group(1) or mm.group(2)
                if tgt in labels and labels[tgt] < i:
                    span = i - labels[tgt]
                    if best is None or span > best[0]:
                        best = (span, labels[tgt], i)
        _, a, b = best
        loop = [l for l in lines[a:b + 1] if not l.endswith(':')]
        c = Counter()
        for l in loop:
            op = l.split()[0]
            if 'f64' in op:
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
        print(f"{name}: loop {len(loop)} instrs | f64 {tot('f64')} v32 {tot('v32')} salu {tot('s:')} mem {tot('mem')} lds {tot('lds')}")
        for v, k in sorted(((v, k) for k, v in c.items()), reverse=True)[:22]:
            print(f"      {v:5d} {k}")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
