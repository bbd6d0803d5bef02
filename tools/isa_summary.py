"""Per-kernel register / spill / scratch / LDS summary of a hipcc -S --cuda-device-only listing (gfx950).
usage: python tools/isa_summary.py file.s"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
md = txt[txt.find('amdhsa.kernels'):]
for it in md.split('  - .agpr_count')[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", it).group(1)
    name = g("name")
    try:
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    print("vgpr %3s sgpr %3s spill %3s scratch %4s lds %6s  %s" % (g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                                                                    g("private_segment_fixed_size"), g("group_segment_fixed_size"), name[:150]))
