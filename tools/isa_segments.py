"""Developer tool: static instruction mix of one kernel in a `hipcc -S` listing, split
at its barriers (the phases of the fused kernels are separated by them), with the
backward branches (loops) marked so that the counts can be weighted by hand.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s
    python tools/isa_segments.py x.s fwd_head_loss_kernelILi256ELi1ELi8ELb1ELi5
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines)
                 if key in l and l.rstrip().split(';')[0].rstrip().endswith(':'))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    labels = {}
    for i in range(start, end):
        m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
        if m:
            labels[m.group(1)] = i
    seg = dict(valu=0, mfma=0, trans=0, lds=0, vmem=0, salu=0, pk=0)
    segs = []
    first = start

    def flush(i, why):
        nonlocal seg, first
        segs.append((first - start, i - start, why, seg))
        seg = dict(valu=0, mfma=0, trans=0, lds=0, vmem=0, salu=0, pk=0)
        first = i

    for i in range(start, end):
        l = lines[i].strip()
        op = l.split()[0] if l else ''
        if op.startswith('v_mfma'):
            seg['mfma'] += 1
        elif op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')):
            seg['trans'] += 1
        elif op.startswith('v_pk_'):
            seg['pk'] += 1
        elif op.startswith('v_'):
            seg['valu'] += 1
        elif op.startswith('ds_'):
            seg['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            seg['vmem'] += 1
        elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_barrier')):
            seg['salu'] += 1
        if op == 's_barrier':
            flush(i, 'barrier')
        m = re.match(r'^s_cbranch\S*\s+(\.LBB\d+_\d+)', l)
        if m and labels.get(m.group(1), end) <= i:
            flush(i, 'LOOP back to line %d' % (labels[m.group(1)] - start))
    flush(end, 'end')
    print('%7s %7s  %5s %4s %5s %5s %4s %5s %5s  %s' % (
        'from', 'to', 'valu', 'pk', 'trans', 'mfma', 'lds', 'vmem', 'salu', 'ends with'))
    for a, b, why, s in segs:
        print('%7d %7d  %5d %4d %5d %5d %4d %5d %5d  %s' % (
            a, b, s['valu'], s['pk'], s['trans'], s['mfma'], s['lds'], s['vmem'],
            s['salu'], why))


if __name__ == '__main__':
    main()
