#!/usr/bin/env python3
"""Turns the counter summary printed by tools/pmc_attn.sh / tools/pmc_sim.sh into the record bench.py reads for
roofline.traffic (profiles/pmc_<class>.json): HBM bytes per launch = 2 x FETCH_SIZE (gfx950 tallies a wide coalesced read
at half its bytes, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB, from their separate passes; stamped with the hash of
everything the kernel is compiled from (bench.kernel_source_hash) and with the batch / shape the pass was taken at.

    python tools/pmc_json.py <class> <summary.txt> <kernel name substring> <out.json> key=value ...   (batch=..., tokens=..., ...)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def main():
    cls, summary, needle, out = sys.argv[1:5]
    extra = {}
    for kv in sys.argv[5:]:
        k, v = kv.split('=', 1)
        extra[k] = int(v) if re.fullmatch(r'-?\d+', v) else v
    counters, cur = {}, None
    for line in open(summary):
        m = re.match(r'\s+(\w+)\s+mean\s+([0-9.eE+-]+)\s+n=(\d+)', line)
        if m and cur is not None:
            counters[cur][m.group(1)] = float(m.group(2))
        elif line.strip() and not line.startswith(' ') and 'rc=' not in line:
            cur = line.strip()
            counters.setdefault(cur, {})
    hits = [k for k in counters if needle in k and 'FETCH_SIZE' in counters[k]]
    if len(hits) != 1:
        raise SystemExit(f'kernel {needle!r}: {len(hits)} matches with FETCH_SIZE among {list(counters)}')
    c = counters[hits[0]]
    rec = {'kernel_trace_name': hits[0], 'hbm_bytes_per_launch': int((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024),
           'fetch_size_kib': c['FETCH_SIZE'], 'write_size_kib': c['WRITE_SIZE'],
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (--kernel-trace only); FETCH_SIZE doubled '
                   '(gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md)',
           'source_sha1': bench.kernel_source_hash(cls), 'sources': list(bench.KERNEL_SOURCES[cls])}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8
        rec['cycles_per_launch'] = int(cyc)
        rec['mfma_pipe_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), 4)
        if 'SQ_ACTIVE_INST_VALU' in c:
            rec['valu_issue_busy'] = round(4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc), 4)
    rec.update(extra)
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
