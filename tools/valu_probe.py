"""VALU issue rates of this box (pathed_hip_measure_valu_modes): G wave-instructions / s per instruction mix and occupancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathed_amd.integrator import VALU_MODES, measure_valu_modes
print("%-42s" % "waves per SIMD" + "".join("%9d" % w for w in (1, 2, 3, 4, 6, 8)))
table = [measure_valu_modes(w) for w in (1, 2, 3, 4, 6, 8)]
for mode, name in enumerate(VALU_MODES):
    print("%-42s" % name + "".join("%9.0f" % (row[mode] / 1e9) for row in table))
