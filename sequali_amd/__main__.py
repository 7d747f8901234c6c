"""python -m sequali_amd reads.fastq[.gz]|reads.bam [mates.fastq[.gz]] [--json out.json] [--raw]

Writes the reference's JSON report of the QC modules (sequali_amd.report: the key set of
report_modules.py:2411-2428 as far as it is restated); --raw: every getter of every module
instead (driver.raw_outputs)."""
import argparse
import json
import sys

from . import driver


def main() -> None:
    ap = argparse.ArgumentParser(prog="python -m sequali_amd", description=driver.__doc__.splitlines()[0])
    ap.add_argument("input")
    ap.add_argument("input_reverse", nargs="?")
    ap.add_argument("--json", help="write the JSON here (default: stdout)")
    ap.add_argument("--raw", action="store_true", help="the getters of the modules instead of the report modules")
    ap.add_argument("--overrepresentation-max-unique-fragments", type=int, default=driver.DEFAULT_MAX_UNIQUE_FRAGMENTS)
    ap.add_argument("--overrepresentation-fragment-length", type=int, default=driver.DEFAULT_FRAGMENT_LENGTH)
    ap.add_argument("--overrepresentation-sample-every", type=int, default=driver.DEFAULT_UNIQUE_SAMPLE_EVERY)
    ap.add_argument("--duplication-max-stored-fingerprints", type=int,
                    default=driver.DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS)
    ap.add_argument("--fingerprint-front-length", type=int, default=driver.DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH)
    ap.add_argument("--fingerprint-back-length", type=int, default=driver.DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH)
    ap.add_argument("--fingerprint-front-offset", type=int)
    ap.add_argument("--fingerprint-back-offset", type=int)
    args = ap.parse_args()
    modules = driver.run(args.input, args.input_reverse,
                         overrepresentation_max_unique_fragments=args.overrepresentation_max_unique_fragments,
                         overrepresentation_fragment_length=args.overrepresentation_fragment_length,
                         overrepresentation_sample_every=args.overrepresentation_sample_every,
                         duplication_max_stored_fingerprints=args.duplication_max_stored_fingerprints,
                         fingerprint_front_length=args.fingerprint_front_length,
                         fingerprint_back_length=args.fingerprint_back_length,
                         fingerprint_front_offset=args.fingerprint_front_offset,
                         fingerprint_back_offset=args.fingerprint_back_offset)
    if args.raw:
        text = json.dumps(driver.raw_outputs(modules))
    else:
        from . import report
        text = json.dumps(report.report(modules, args.input, args.input_reverse))
    if args.json:
        with open(args.json, "wt") as f:
            f.write(text)
    else:
        sys.stdout.write(text + "\n")


if __name__ == "__main__":
    main()
