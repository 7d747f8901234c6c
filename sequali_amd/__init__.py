"""sequali_amd -- MI355X-native per-read QC accumulators behind the API of
sequali's ``_qc`` extension (QCMetrics, AdapterCounter, PerTileQuality,
OverrepresentedSequences, DedupEstimator, InsertSizeMetrics and the
FastqRecordView / FastqRecordArrayView / FastqParser boundary types).

    from sequali_amd import QCMetrics, FastqParser
    metrics = QCMetrics()
    for record_array in FastqParser(open("reads.fastq", "rb"), 64 * 1024 * 1024):
        metrics.add_record_array(record_array)
    table = metrics.base_count_table()
"""
from ._qc import (A, C_ as C, G, N, T, DEFAULT_BASES_FROM_END, DEFAULT_BASES_FROM_START,
                  DEFAULT_DEDUP_MAX_STORED_FINGERPRINTS, DEFAULT_END_ANCHOR_LENGTH,
                  DEFAULT_FINGERPRINT_BACK_SEQUENCE_LENGTH,
                  DEFAULT_FINGERPRINT_BACK_SEQUENCE_OFFSET,
                  DEFAULT_FINGERPRINT_FRONT_SEQUENCE_LENGTH,
                  DEFAULT_FINGERPRINT_FRONT_SEQUENCE_OFFSET, DEFAULT_FRAGMENT_LENGTH,
                  DEFAULT_MAX_UNIQUE_FRAGMENTS, DEFAULT_UNIQUE_SAMPLE_EVERY,
                  INSERT_SIZE_MAX_ADAPTER_STORE_SIZE, MAX_SEQUENCE_SIZE, NUMBER_OF_NUCS,
                  NUMBER_OF_PHREDS, PHRED_MAX, TABLE_SIZE, AdapterCounter, BamParser, DedupEstimator,
                  FastqParser, FastqRecordArrayView, FastqRecordView, FusedPass,
                  InsertSizeMetrics, NanoporeReadInfo, NanoStats, OverrepresentedSequences,
                  PairedPass, PerTileQuality, PinnedReader, QCMetrics)

__all__ = [
    "A", "C", "G", "N", "T", "AdapterCounter", "BamParser", "DedupEstimator", "FastqParser",
    "FastqRecordArrayView", "FastqRecordView", "FusedPass", "InsertSizeMetrics",
    "NanoStats", "NanoporeReadInfo",
    "OverrepresentedSequences", "PairedPass", "PerTileQuality", "PinnedReader", "QCMetrics", "NUMBER_OF_NUCS",
    "NUMBER_OF_PHREDS", "PHRED_MAX", "TABLE_SIZE", "MAX_SEQUENCE_SIZE",
]
