"""Text summary of the matrix-core counters a default `python bench.py` run collected about its own contraction launches
(`roofline.mfma_pmc` of the JSON line):   python tools/pmc_mfma_summary.py profiles/r06_bf16_bench.json [session-name] > profiles/r06_bf16_contraction_pmc.txt"""
import json
import sys


def main():
    j = json.load(open(sys.argv[1]))
    session = sys.argv[2] if len(sys.argv) > 2 else '?'
    r = j['roofline']
    m = r['mfma_pmc']
    clk = m['clock_ghz']
    out = ["# MFMA utilisation of the relation contractions of config 3 (bf16), measured INSIDE the default `python bench.py` run of the",
           "# round's final tree (session %s; bench.py: pmc_mfma_in_run -- a rocprofv3 child of the same bench.py, --kernel-trace --pmc," % session,
           "# 2 warm-up + 1 counted iterations; the counters below are per launch, mean over the launches of each kernel).",
           "# source: " + m['source'],
           "#",
           "# busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)        share of SIMD cycles with the matrix pipe busy",
           "# clock  = GRBM_GUI_ACTIVE / 8 / launch time in the same pass                        what the power management granted",
           "# parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                              waves at s_waitcnt / s_barrier",
           "# TFLOP/s = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 flop / launch time (cross-check of the flop count: executed, incl. tile padding)",
           "",
           "%-58s %8s %10s %8s %8s %8s %10s" % ('kernel', 'launches', 'time ms', 'busy', 'clock', 'parked', 'TFLOP/s')]

    def val(c, k):
        return c[k]['per_launch'] if isinstance(c[k], dict) else c[k]
    for name, c in m['per_kernel'].items():
        cyc = val(c, 'GRBM_GUI_ACTIVE') / 8.0
        t = cyc / (clk * 1e9) * 1e3
        out.append("%-58s %8d %10.3f %8.3f %8.2f %8.3f %10.1f"
                   % (name[:58], c['GRBM_GUI_ACTIVE']['launches'], t, val(c, 'SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024), clk,
                      val(c, 'SQ_WAIT_ANY') / val(c, 'SQ_WAVE_CYCLES'), val(c, 'SQ_INSTS_VALU_MFMA_MOPS_BF16') * 512 / (t * 1e-3) / 1e12))
    t = m['avg_launch_ms_under_counters']
    out.append("%-58s %8d %10.3f %8.3f %8.2f %8.3f %10.1f" % ('all contraction launches', m['launches_counted'], t, m['mfma_busy'], clk,
                                                           m['waves_parked'], m['mfma_tflops_at_clock']))
    out += ["",
            "# beside it, the hipEvent figure of the timed region of the same run: %.1f TFLOP/s algorithmic = %.3f of the 2.5 PFLOP/s dense bf16 peak"
            % (r['achieved'], r['frac']),
            "# (%.3f ms per launch; under the counters %.3f ms).  At the granted clock the peak is 2.5 x %.2f / 2.4 = %.2f PFLOP/s: the launches run at"
            % (r['avg_launch_ms'], t, clk, 2.5 * clk / 2.4),
            "# %.2f of THAT, which is what `busy` says.  HBM beside it: %.2f GB per launch from the counters = %.2f x the algorithmic bytes (two passes over"
            % (r['achieved'] / 1000 / (2.5 * clk / 2.4), r['traffic'] / 1e9, r['traffic_ratio']),
            "# every relation), %.2f of 8 TB/s as scheduled.  Round 2's stand-alone figure was 0.61 busy (profiles/r02_bf16_contraction_pmc.txt) on"
            % r['hbm_scheduled']['frac'],
            "# the stand-alone contraction at a higher clock; inside the iteration the second stream's products share the chip."]
    print('\n'.join(out))


if __name__ == '__main__':
    main()
