"""VALU issue rates of this box (pathed_hip_measure_valu_modes): G wave-instructions / s per instruction mix and occupancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathed_amd.integrator import VALU_MODES, measure_valu_modes
print("%-42s" % "waves per SIMD" + "".join("%9d" % w for w in (1, 2, 3, 4, 6, 8)))
table = [measure_valu_modes(w) for w in (1, 2, 3, 4, 6, 8)]
for mode, name in enumerate(VALU_MODES):
    print("%-42s" % name + "".join("%9.0f" % (row[mode] / 1e9) for row in table))

# the probe with its own clocks: cycles per instruction at the frequency the chip actually ran at
from pathed_amd.integrator import measure_valu_clocks
print()
print("v_fma_f32, three VGPR operands, with the wave's own clocks (s_memtime / s_memrealtime):")
print("%6s %7s %12s %12s %14s %16s %18s" % ("waves", "chains", "G instr/s", "shader MHz", "ticks/instr", "cycles/instr", "cycles/instr (ev)"))
for waves in (1, 2, 4, 8):
    for chains in (8, 16):
        p = measure_valu_clocks(waves, chains)
        print("%6d %7d %12.1f %12.1f %14.3f %16.3f %18.3f" % (waves, chains, p["rate"] / 1e9, p["shader_clock_mhz"], p["wave_ticks_per_instruction"],
                                                        p["cycles_per_instruction"], p["cycles_per_instruction_events"]))
print("wall clock %.1f MHz, advertised peak clock %.1f MHz; guide: 2 cycles per wave64 v_fma_f32 = 1228.8 G/s at 2400 MHz on 1024 SIMDs" % (
    p["wall_clock_mhz"], p["peak_clock_mhz"]))
