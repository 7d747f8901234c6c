"""cProfile of the reference's call pattern from host text: FastqParser at its default buffer size
feeding FusedPass(QCMetrics, AdapterCounter).  python scripts/prof_e2e.py [reads]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sequali_amd import AdapterCounter, FastqParser, FusedPass, QCMetrics, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
text = synth.illumina_fastq(0, n)
def run():
    f = FusedPass(QCMetrics(), AdapterCounter(list(synth.ILLUMINA_PROBES)))
    for a in FastqParser(io.BytesIO(text)):
        f.add_record_array(a)
    return f.qc_metrics.base_count_table()
run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
print(f"{n * 150 / dt / 1e9:.3f} Gbases/s, {dt:.3f} s")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
