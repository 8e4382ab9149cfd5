"""Developer tool: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/*.json.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> \\
        <write_counter_collection.csv> profiles/r01

Applies the gfx950 corrections of MI355X_MICROARCH.md (HBM / rocprofv3 section):
counters are KiB; FETCH_SIZE reports half of wide coalesced reads, so x2.
Bytes are averages per launch over every launch of a kernel name.
"""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    depth = 0
    for i, ch in enumerate(name):  # drop the parameter list
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return name[:i].strip()
    return name.strip()


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short(r['Kernel_Name'])
        acc[k][0] += float(r['Counter_Value'])
        acc[k][1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    prefix = sys.argv[3]
    raw, out = {}, {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        n = max(nf, nw, 1)
        raw[k] = dict(launches=n, FETCH_SIZE_KiB=f / max(nf, 1),
                      WRITE_SIZE_KiB=w / max(nw, 1))
        fb = 2.0 * 1024.0 * f / max(nf, 1)
        wb = 1024.0 * w / max(nw, 1)
        out[k] = dict(launches=n, fetch_bytes_corrected=fb, write_bytes=wb,
                      hbm_bytes=fb + wb)
    json.dump(dict(counters='per-launch averages, KiB as reported', kernels=raw),
              open(prefix + '_pmc_traffic_raw.json', 'w'), indent=1)
    json.dump(dict(
        method='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
               'over `python3 bench.py --steps 1 --warmup 0 --cpu-envs 0 '
               '--no-roofline --no-overlap` (C3); counters are in KiB; FETCH_SIZE '
               'doubled (gfx950 reports half of wide coalesced reads, '
               'MI355X_MICROARCH.md HBM section); bytes are averages per launch '
               'over every launch of the kernel in one iteration',
        kernels=out), open(prefix + '_traffic.json', 'w'), indent=1)
    for k, v in out.items():
        if 'gemm' in k or 'scan' in k or 'skinny' in k:
            print('%-70s n=%5d  %.2f MB' % (k[:70], v['launches'], v['hbm_bytes'] / 1e6))


if __name__ == '__main__':
    main()
