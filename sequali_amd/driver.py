"""The reference's driver loop over the GPU modules.

Restates ``sequali.__main__.main`` (__main__.py:199-306) up to, not including, the report:
opens one FASTQ / uBAM file or a FASTQ pair (plain, gzip or BGZF), sniffs the sequencing
technology (util.py:162-253), picks the adapters for it (adapters.py:32-48 over the list
below), feeds every record array to the modules in the reference's order and returns the
raw module outputs -- the getter values ``calculate_stats`` (report_modules.py:2607) starts
from.  The reference's report JSON is derived from exactly these values; it is not
reproduced here (report_modules needs pygal, which this image lacks, so there is nothing to
pin it against).

    python -m sequali_amd reads.fastq.gz [mates.fastq.gz] [--json out.json]
"""
from __future__ import annotations

import io
import re
import zlib
from typing import Dict, List, NamedTuple, Optional

from ._qc import (DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS, DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH,
                  DEFAULT_FINGERPRINT_BACK_SEQUENCE_OFFSET, DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH,
                  DEFAULT_FINGERPRINT_FRONT_SEQUENCE_OFFSET, DEFAULT_FRAGMENT_LENGTH,
                  DEFAULT_MAX_UNIQUE_FRAGMENTS, DEFAULT_UNIQUE_SAMPLE_EVERY, AdapterCounter, BamParser,
                  DedupEstimator, FastqParser, FusedPass, InsertSizeMetrics, NanoStats, PairedPass,
                  OverrepresentedSequences, PerTileQuality, QCMetrics)

# __main__.py:40-41
DEFAULT_FINGERPRINT_FRONT_SEQUENCE_PAIRED_OFFSET = 0
DEFAULT_FINGERPRINT_BACK_SEQUENCE_PAIRED_OFFSET = 0


class Adapter(NamedTuple):
    name: str
    sequencing_technology: str
    sequence: str
    sequence_position: str


# data of the reference's adapters/adapter_list.tsv (name, technology, sequence, position)
ADAPTERS = [Adapter(*row) for row in (
    ("Illumina Universal Adapter", "illumina", "AGATCGGAAGAG", "end"),
    ("Illumina Small RNA 3' adapter", "illumina", "TGGAATTCTCGG", "end"),
    ("Illumina Small RNA 5' adapter", "illumina", "GATCGTCGGACT", "end"),
    ("Nextera Transposase Sequence", "illumina", "CTGTCTCTTATA", "end"),
    ("PolyG", "illumina", "GGGGGGGGGGGG", "end"),
    ("PolyA", "illumina", "AAAAAAAAAAAA", "end"),
    ("Oxford nanopore ligation kit or Adapter Mix (AMX), top strand", "nanopore", "TTACGTATTGCT", "start"),
    ("Oxford nanopore ligation kit or Adapter Mix (AMX), bottom strand ", "nanopore", "GCAATACGTAAC", "end"),
    ("Oxford nanopore cDNA RT Adapter (CRT)", "nanopore", "CTTGCGGGCGGC", "end"),
    ("Oxford nanopore RT Adapter (RTA), top strand", "nanopore", "GGTAGTAGGTTC", "start"),
    ("Oxford nanopore RT Adapter (RTA), and RNA Adapter Mix (RMX), bottom strand", "nanopore", "GAGGCGAGCGGT", "end"),
    ("Oxford nanopore RNA Adapter Mix (RMX); top strand", "nanopore", "CAAGATACGCAC", "start"),
    ("Oxford nanpore cDNA primer, forward sequence", "nanopore", "GTGACTTGCCTG", "start"),
    ("Oxford nanopore CDNA primer, forward and reverse sequence", "nanopore", "ATCGCCTACCGT", "end"),
    ("Oxford nanopore VN primer", "nanopore", "TCTATCTTCTTT", "end"),
    ("Oxford nanopore RT Primer (RTP)", "nanopore", "TCTTCAGAGGAG", "start"),
    ("Oford nanopore Strand Switching Primer (SSP)", "nanopore", "GATATTGCTGGG", "start"),
    ("Oxford nanopore Strand Switching Primer II (SSPII)", "nanopore", "TGATATTGCTTT", "start"),
    ("Oxford nanopore Native Adapter (NA), top strand", "nanopore", "GTACGTATTGCT", "start"),
    ("Oxford nanopore Native Adapter (NA), bottom strand", "nanopore", "ACGTAACTGAAC", "end"),
)]


def adapters_for(sequencing_technology: Optional[str]) -> List[Adapter]:
    """adapters_from_file, adapters.py:32-48: every adapter when the technology is unknown"""
    return [a for a in ADAPTERS if sequencing_technology is None or
            a.sequencing_technology in (sequencing_technology, "all")]


_UUID_NAME = re.compile(r"[0-9a-fA-F]{8}(-[0-9a-fA-F]{4}){3}-[0-9a-fA-F]{12}\Z")


def fastq_header_is_illumina(header: str) -> bool:
    """Same verdict as util.py:187-210: `<instrument>:<run>:<flowcell>:<lane>:<tile>:<x>:<y>`
    (six colons in the read name) and, when a comment follows, `<read>:<Y|N>:<control>:<sample>`."""
    tokens = header.split(maxsplit=1)
    if not tokens:
        return False
    name = tokens[0]
    comment = tokens[1] if len(tokens) > 1 else ""
    if comment:
        fields = comment.split(":")
        if len(fields) != 4 or fields[1] not in ("Y", "N"):
            return False
    return name.count(":") == 6


def fastq_header_is_nanopore(header: str) -> bool:
    """Same verdict as util.py:213-235: a UUID read name followed by key=value fields among
    which one starts with `ch` and one with `st` (guppy FASTQ, or uBAM converted to FASTQ)."""
    tokens = header.split()
    if not tokens or not _UUID_NAME.match(tokens[0]):
        return False
    keys = tokens[1:]
    return any(k[:2] == "ch" for k in keys) and any(k[:2] == "st" for k in keys)


def technology_from_bam_header(header: bytes) -> Optional[str]:
    """Same verdict as util.py:238-253: the platform (PL) of the first read group that names
    ONT or Illumina."""
    platforms = {"ONT": "nanopore", "Illumina": "illumina"}
    for line in header.decode("utf-8").splitlines():
        if not line.startswith("@RG"):
            continue
        for field in line.split("\t")[1:]:
            tag, value = field.split(":", maxsplit=1)
            if tag == "PL" and value in platforms:
                return platforms[value]
    return None


class _BgzfReader(io.RawIOBase):
    """the concatenated gzip members of a BGZF file as one stream"""

    def __init__(self, raw: io.BufferedIOBase):
        self._raw, self._z, self._buf = raw, zlib.decompressobj(31), b""

    def readable(self) -> bool:
        return True

    def readinto(self, out) -> int:
        while not self._buf:
            if self._z.eof:
                rest = self._z.unused_data
                self._z = zlib.decompressobj(31)
                if rest:
                    self._buf = self._z.decompress(rest)
                    continue
            data = self._raw.read(1 << 16)
            if not data:
                return 0
            self._buf = self._z.decompress(data)
        n = min(len(out), len(self._buf))
        out[:n] = self._buf[:n]
        self._buf = self._buf[n:]
        return n


class NGSFile:
    """util.py NGSFile: a FASTQ or uBAM file, compressed or not, as an iterator of record arrays"""

    def __init__(self, path: str, buffersize: int = 64 << 20, split_on_device: bool = True):
        self.filepath = path
        raw = open(path, "rb")
        if raw.peek(2)[:2] == b"\x1f\x8b":
            stream = io.BufferedReader(_BgzfReader(raw), 1 << 20)
        else:
            stream = raw
        self._stream = stream
        head = stream.peek(io.DEFAULT_BUFFER_SIZE)
        if head[:4] == b"BAM\x01":
            self.format = "BAM"
            self._reader = BamParser(stream, buffersize)
            self.sequencing_technology = technology_from_bam_header(self._reader.header)
        else:
            self.format = "FASTQ"
            self.sequencing_technology = None
            if head[:1] == b"@":
                end = head.find(b"\n")
                header = head[1:end if end != -1 else None].decode("ascii")
                if fastq_header_is_illumina(header):
                    self.sequencing_technology = "illumina"
                elif fastq_header_is_nanopore(header):
                    self.sequencing_technology = "nanopore"
            self._reader = FastqParser(stream, buffersize, split_on_device=split_on_device)
            self._host_reader = None

    def __iter__(self):
        return iter(self._reader)

    def read(self, number_of_records: int):
        return self._reader.read(number_of_records)

    def close(self):
        self._stream.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def sequence_names_match(name1: str, name2: str) -> bool:
    """Same verdict as util.py:256-268: the read ids (up to the first blank) are equal, a
    trailing 1 / 2 mate digit aside."""
    ids = [n.split(maxsplit=1)[0] for n in (name1, name2)]
    if {ids[0][-1], ids[1][-1]} == {"1", "2"}:
        ids = [i[:-1] for i in ids]
    return ids[0] == ids[1]


def run(input_path: str, input_reverse: Optional[str] = None, *,
        overrepresentation_max_unique_fragments: int = DEFAULT_MAX_UNIQUE_FRAGMENTS,
        overrepresentation_fragment_length: int = DEFAULT_FRAGMENT_LENGTH,
        overrepresentation_sample_every: int = DEFAULT_UNIQUE_SAMPLE_EVERY,
        duplication_max_stored_fingerprints: int = DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS,
        fingerprint_front_length: int = DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH,
        fingerprint_back_length: int = DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH,
        fingerprint_front_offset: Optional[int] = None, fingerprint_back_offset: Optional[int] = None,
        buffersize: int = 64 << 20) -> Dict[str, object]:
    """__main__.py:199-306: the module objects after every record array went through them"""
    paired = bool(input_reverse)
    overrep_kw = dict(max_unique_fragments=overrepresentation_max_unique_fragments,
                      fragment_length=overrepresentation_fragment_length,
                      sample_every=overrepresentation_sample_every)
    metrics1, per_tile1, nanostats1 = QCMetrics(), PerTileQuality(), NanoStats()
    overrep1 = OverrepresentedSequences(**overrep_kw)
    if fingerprint_front_offset is None:  # :226-241
        fingerprint_front_offset = (DEFAULT_FINGERPRINT_FRONT_SEQUENCE_PAIRED_OFFSET if paired
                                    else DEFAULT_FINGERPRINT_FRONT_SEQUENCE_OFFSET)
    if fingerprint_back_offset is None:
        fingerprint_back_offset = DEFAULT_FINGERPRINT_BACK_SEQUENCE_PAIRED_OFFSET
    dedup = DedupEstimator(max_stored_fingerprints=duplication_max_stored_fingerprints,
                           front_sequence_length=fingerprint_front_length,
                           front_sequence_offset=fingerprint_front_offset,
                           back_sequence_length=fingerprint_back_length,
                           back_sequence_offset=fingerprint_back_offset)
    insert_sizes = metrics2 = per_tile2 = overrep2 = adapter_counter1 = None
    if paired:
        insert_sizes, metrics2, per_tile2 = InsertSizeMetrics(), QCMetrics(), PerTileQuality()
        overrep2 = OverrepresentedSequences(**overrep_kw)
    with NGSFile(input_path, buffersize, split_on_device=not paired) as reader1:
        seqtech = reader1.sequencing_technology
        reader2 = None
        if paired:
            reader2 = NGSFile(input_reverse, buffersize, split_on_device=False)
            if reader1.sequencing_technology != reader2.sequencing_technology:
                raise RuntimeError(f"Mismatching sequencing technologies:\n"
                                   f"{reader1.filepath}: {reader1.sequencing_technology}\n"
                                   f"{reader2.filepath}: {reader2.sequencing_technology}\n")
            if not (reader1.format == "FASTQ" and reader2.format == "FASTQ"):
                raise RuntimeError("Paired end mode is only supported for FASTQ files.")
            seqtech = "illumina"  # :277
        adapters = adapters_for(seqtech)
        if not paired:
            adapter_counter1 = AdapterCounter(a.sequence for a in adapters)
        # one read of each array from HBM for the three per-base modules (same tables as
        # the three separate calls of the reference's loop)
        fused1 = FusedPass(metrics1, adapter_counter1, per_tile1) if not paired else None
        # paired input: the five calls the reference makes per pair of arrays (metrics, per tile quality on both mates,
        # insert sizes: __main__.py:279-306) as one (sq_paired_add_batches; csrc/sq_pair.hip)
        paired_pass = PairedPass(metrics1, per_tile1, metrics2, per_tile2, insert_sizes) if paired else None
        try:
            for arr1 in reader1:
                if not paired:
                    fused1.add_record_array(arr1)
                    overrep1.add_record_array(arr1)
                    nanostats1.add_record_array(arr1)
                if paired:
                    arr2 = reader2.read(len(arr1))
                    if len(arr1) != len(arr2):
                        raise RuntimeError(f"FASTQ Files out of sync {input_path} has more FASTQ records "
                                           f"than {input_reverse}.")
                    if not arr1.is_mate(arr2):
                        for i in range(len(arr1)):
                            n1, n2 = arr1[i].name(), arr2[i].name()
                            if not sequence_names_match(n1, n2):
                                raise RuntimeError(f"Mismatching names found! {n1} {n2}")
                        raise RuntimeError("Mismatching names found!")
                    paired_pass.add_record_array_pair(arr1, arr2)
                    overrep1.add_record_array(arr1)
                    nanostats1.add_record_array(arr1)      # behind QCMetrics' pass over arr1: it reads accumulated_error_rate (:5314)
                    dedup.add_record_array_pair(arr1, arr2)
                    overrep2.add_record_array(arr2)
                else:
                    dedup.add_record_array(arr1)
            if paired and len(reader2.read(1)) > 0:
                raise RuntimeError(f"FASTQ Files out of sync {input_reverse} has more FASTQ records "
                                   f"than {input_path}.")
        finally:
            if reader2 is not None:
                reader2.close()
    metrics1.flush()
    return dict(metrics=metrics1, adapter_counter=adapter_counter1, per_tile_quality=per_tile1,
                sequence_duplication=overrep1, dedup_estimator=dedup, nanostats=nanostats1,
                insert_size_metrics=insert_sizes, metrics_reverse=metrics2,
                per_tile_quality_reverse=per_tile2, sequence_duplication_reverse=overrep2,
                adapters=adapters, sequencing_technology=seqtech)


def raw_outputs(modules: Dict[str, object]) -> Dict[str, object]:
    """every getter of every module as plain lists / numbers (JSON serialisable)"""
    out: Dict[str, object] = {"sequencing_technology": modules["sequencing_technology"],
                              "adapters": [a._asdict() for a in modules["adapters"]]}

    def qc(m):
        return dict(number_of_reads=m.number_of_reads, max_length=m.max_length,
                    base_count_table=list(m.base_count_table()), phred_count_table=list(m.phred_count_table()),
                    end_anchored_base_count_table=list(m.end_anchored_base_count_table()),
                    end_anchored_phred_count_table=list(m.end_anchored_phred_count_table()),
                    gc_content=list(m.gc_content()), phred_scores=list(m.phred_scores()))

    def tiles(p):
        return dict(number_of_reads=p.number_of_reads, max_length=p.max_length, skipped_reason=p.skipped_reason,
                    tile_counts=[[t, list(e), list(c)] for t, e, c in p.get_tile_counts()])

    def overrep(o):
        return dict(number_of_sequences=o.number_of_sequences, sampled_sequences=o.sampled_sequences,
                    collected_unique_fragments=o.collected_unique_fragments, total_fragments=o.total_fragments,
                    overrepresented_sequences=[list(x) for x in o.overrepresented_sequences()])

    out["metrics"] = qc(modules["metrics"])
    out["per_tile_quality"] = tiles(modules["per_tile_quality"])
    out["sequence_duplication"] = overrep(modules["sequence_duplication"])
    d = modules["dedup_estimator"]
    out["dedup_estimator"] = dict(tracked_sequences=d.tracked_sequences, modulo_bits=d._modulo_bits,
                                  duplication_counts=list(d.duplication_counts()))
    ns = modules["nanostats"]
    out["nanostats"] = dict(number_of_reads=ns.number_of_reads, skipped_reason=ns.skipped_reason,
                            minimum_time=ns.minimum_time, maximum_time=ns.maximum_time,
                            nano_infos=[[i.start_time, i.channel_id, i.length, i.cumulative_error_rate, i.duration,
                                         i.parent_id_hash] for i in ns.nano_info_iterator()])
    if modules["adapter_counter"] is not None:
        a = modules["adapter_counter"]
        out["adapter_counter"] = dict(number_of_sequences=a.number_of_sequences, max_length=a.max_length,
                                      counts=[[s, list(f), list(r)] for s, f, r in a.get_counts()])
    if modules["insert_size_metrics"] is not None:
        z = modules["insert_size_metrics"]
        out["insert_size_metrics"] = dict(total_reads=z.total_reads, insert_sizes=list(z.insert_sizes()),
                                          number_of_adapters_read1=z.number_of_adapters_read1,
                                          number_of_adapters_read2=z.number_of_adapters_read2,
                                          adapters_read1=[list(x) for x in z.adapters_read1()],
                                          adapters_read2=[list(x) for x in z.adapters_read2()])
        out["metrics_reverse"] = qc(modules["metrics_reverse"])
        out["per_tile_quality_reverse"] = tiles(modules["per_tile_quality_reverse"])
        out["sequence_duplication_reverse"] = overrep(modules["sequence_duplication_reverse"])
    return out
