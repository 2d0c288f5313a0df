"""Summarise `make resources` output (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else 'resource_usage.txt').read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split('\n')[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else '?'
    short = name[:100]
    print("%-100s vgpr=%4s agpr=%4s spill=%s scratch=%s occ=%s lds=%s" % (
        short, g('VGPRs'), g('AGPRs'), g('VGPR Spill'), g(r'ScratchSize \[bytes/lane\]'),
        g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
