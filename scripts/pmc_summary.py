#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc --output-format csv run: scripts/pmc_summary.py <dir> <kernel-substring>."""
import csv, glob, os, sys
from collections import defaultdict
d, kern = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
if not files:
    print('no counter_collection.csv under', d, [f for f in glob.glob(os.path.join(d, '**', '*'), recursive=True)][:20])
    sys.exit(0)
tot = defaultdict(float); cnt = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')
        if kern in k:
            short = k[:70]
            tot[(short, row['Counter_Name'])] += float(row['Counter_Value'])
            cnt[short].add(row.get('Dispatch_Id', ''))
for (k, c), v in sorted(tot.items()):
    print('%s dispatches=%d %s = %.6g' % (k, len(cnt[k]), c, v))
