"""Folds `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile) into one line per kernel:
   python tools/rpass_summary.py build/ks_probe.rpass   (development aid)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip().split()[0] for b in blocks]
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
for b, dn in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = re.sub(r'\(.*', '', dn).replace('void hefx::', '')
    print("%-50s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %s" % (dn, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
