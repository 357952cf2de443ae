"""Resource usage of the kernels whose name contains argv[2], from a -Rpass-analysis=kernel-resource-usage build log (tools/build_variant.sh)."""
import re, sys
log = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "dec_crit_x3"
for b in re.split(r'remark: [^\n]*Function Name: ', log)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    if pat in name:
        g = lambda k: (re.search(k + r': (\d+)', b) or [0, '?'])[1]
        print(name[:80], 'VGPR', g('VGPRs'), 'spill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'SGPR', g('SGPRs'), 'sspill', g('SGPRs Spill'), 'LDS', g(r'LDS Size \[bytes/block\]'))
