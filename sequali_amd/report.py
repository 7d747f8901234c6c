"""The reference's JSON report of the QC modules, as plain data.

`sequali` writes `{module name: module.to_dict()}` for the report modules `calculate_stats` builds
from the hot-path objects (report_modules.py:2607-2682, 2461-2471; __main__.py:313-342).  This
file restates, without the plotting (pygal is not a dependency here), what those modules derive
from the getters, under the reference's key names:

    meta, summary[_read2], sequence_length_distribution[_read2], per_sequence_quality_scores[_read2],
    per_position_base_content[_read2], per_position_n_content[_read2], per_sequence_gc_content[_read2],
    adapter_content, duplication_fractions, overrepresented_sequences[_read2], insert_size_metrics

Not restated: the sequence identification against the contaminant database (out of
scope; the three fields it fills are None), per_position_mean_quality_and_spread,
per_position_quality_distribution, per_tile_quality, nanopore_metrics and
adapter_content_from_overlap (their inputs are in driver.raw_outputs).

Pinned only by the expectations of the reference's integration tests (tests/test_integration.py:
29-42, 97-124, 203-211), re-expressed in tests/test_gpu_driver.py: `report_modules` itself cannot
be imported here (pygal), so no golden vectors exist for this layer -- parity unpinned beyond
those expectations.
"""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional, Sequence, Tuple

NUMBER_OF_NUCS, NUMBER_OF_PHREDS = 5, 12
A, C, G, T, N = 0, 1, 2, 3, 4
READ1, READ2 = "Read 1", "Read 2"
DEFAULT_FRACTION_THRESHOLD, DEFAULT_MIN_THRESHOLD, DEFAULT_MAX_THRESHOLD = 0.0001, 100, (1 << 63) - 1


def equidistant_ranges(length: int, parts: int) -> List[Tuple[int, int]]:
    """report_modules.py:258-269"""
    size, remainder = divmod(length, parts)
    small_parts = parts - remainder
    out, start = [], 0
    for i in range(parts):
        part = size if i < small_parts else size + 1
        if part == 0:
            continue
        out.append((start, start + part))
        start += part
    return out


def logarithmic_ranges(length: int, min_distance: int = 5) -> List[Tuple[int, int]]:
    """report_modules.py:272-290"""
    scaling_factor = 250_000_000 ** (1 / 400)
    out, i, start = [], 0, 0
    while True:
        stop = round(scaling_factor ** i)
        i += 1
        if stop >= start + min_distance:
            out.append((start, stop))
            start = stop
            if stop >= length:
                return out


def stringify_ranges(ranges) -> List[str]:
    return [f"{a + 1}-{b}" if a + 1 != b else f"{a + 1}" for a, b in ranges]


def aggregate_count_matrix(counts: Sequence[int], ranges, table_size: int) -> List[int]:
    """report_modules.py:307-322"""
    out = [0] * (table_size * len(ranges))
    for k, (a, b) in enumerate(ranges):
        for i in range(table_size):
            out[k * table_size + i] = sum(counts[a * table_size + i:b * table_size:table_size])
    return out


def data_ranges_of(max_length: int, graph_resolution: int = 200):
    """calculate_stats, report_modules.py:2626-2630"""
    return logarithmic_ranges(max_length) if max_length > 500 else equidistant_ranges(max_length, graph_resolution)


def summary(metrics, ranges, read_pair_info=None) -> dict:
    """qc_metrics_modules, report_modules.py:2537-2576"""
    base = list(metrics.base_count_table())
    phred = list(metrics.phred_count_table())
    ag_base = aggregate_count_matrix(base, ranges, NUMBER_OF_NUCS)
    ag_phred = aggregate_count_matrix(phred, ranges, NUMBER_OF_PHREDS)
    sum_base = aggregate_count_matrix(ag_base, [(0, len(ag_base) // NUMBER_OF_NUCS)], NUMBER_OF_NUCS)
    sum_phred = aggregate_count_matrix(ag_phred, [(0, len(ag_phred) // NUMBER_OF_PHREDS)], NUMBER_OF_PHREDS)
    total_bases = sum(sum_base)
    total_reads = metrics.number_of_reads
    minimum_length = 0
    for i in range(0, len(base), NUMBER_OF_NUCS):
        if sum(base[i:i + NUMBER_OF_NUCS]) < total_reads:
            break
        minimum_length += 1
    return dict(mean_length=total_bases / max(total_reads, 1), minimum_length=minimum_length,
                maximum_length=metrics.max_length, total_reads=total_reads,
                q20_reads=sum(list(metrics.phred_scores())[20:]), total_bases=total_bases,
                q20_bases=sum(sum_phred[5:]), total_gc_bases=sum_base[C] + sum_base[G],
                total_n_bases=sum_base[N], read_pair_info=read_pair_info)


def sequence_length_distribution(base: Sequence[int], total_sequences: int, ranges, read_pair_info=None) -> dict:
    """SequenceLengthDistribution.from_base_count_tables, report_modules.py:575-636"""
    max_length = len(base) // NUMBER_OF_NUCS
    lengths_at = [0] * (max_length + 1)
    at_least = [0] * (max_length + 1)
    at_least[0] = total_sequences
    for i in range(max_length):
        at_least[i + 1] = sum(base[i * NUMBER_OF_NUCS:(i + 1) * NUMBER_OF_NUCS])
    previous = 0
    for i in range(max_length, 0, -1):
        lengths_at[i] = at_least[i] - previous
        previous = at_least[i]
    counts = [sum(lengths_at[1:][a:b]) for a, b in ranges]
    percentiles = [1, 5, 10, 25, 50, 75, 90, 95, 99]
    thresholds = [int(p * total_sequences / 100) for p in percentiles]
    plen = [0] * len(percentiles)
    ti, accumulated, done = 0, 0, False
    for length, count in enumerate(lengths_at):
        while count > 0 and not done:
            remaining = thresholds[ti] - accumulated
            if count > remaining:
                accumulated += remaining
                plen[ti] = length
                count -= remaining
                ti += 1
                if ti == len(thresholds):
                    done = True
                    break
                continue
            break
        accumulated += count
        if done:
            break
    total_bases = sum(base)
    half, tenth = total_bases // 2, int(total_bases * 0.1)
    sum_bases, n50, n90 = 0, None, None
    for length, number in enumerate(lengths_at):
        sum_bases += length * number
        if n90 is None and sum_bases >= tenth:
            n90 = length
        if n50 is None and sum_bases >= half:
            n50 = length
            break
    keys = ["q1", "q5", "q10", "q25", "q50", "q75", "q90", "q95", "q99"]
    out = dict(length_ranges=["0"] + stringify_ranges(ranges), counts=[lengths_at[0]] + counts)
    out.update(zip(keys, plen))
    out.update(n50=n50, n90=n90, read_pair_info=read_pair_info)
    return out


def base_content_distribution_table(base: Sequence[int]) -> Dict[str, List[float]]:
    """PerPositionBaseContent.base_content_distribution_table, report_modules.py:1142-1166"""
    n = len(base) // NUMBER_OF_NUCS
    frac = [[0.0] * n for _ in range(4)]
    for i in range(n):
        t = base[i * NUMBER_OF_NUCS:(i + 1) * NUMBER_OF_NUCS]
        named = sum(t) - t[N]
        if named == 0:
            continue
        for b in (A, C, G, T):
            frac[b][i] = t[b] / named
    return {"A": frac[A], "C": frac[C], "G": frac[G], "T": frac[T]}


def per_position_base_content(metrics, ranges, read_pair_info=None) -> dict:
    """report_modules.py:1170-1192 with the inputs of qc_metrics_modules (:2541-2552, 2594-2598)"""
    base = list(metrics.base_count_table())
    ag = aggregate_count_matrix(base, ranges, NUMBER_OF_NUCS)
    f = base_content_distribution_table(ag)
    front = base[:metrics.end_anchor_length * NUMBER_OF_NUCS]
    return dict(x_labels=stringify_ranges(ranges), A=f["A"], C=f["C"], G=f["G"], T=f["T"],
                front_anchored=base_content_distribution_table(front),
                end_anchored=base_content_distribution_table(list(metrics.end_anchored_base_count_table())),
                read_pair_info=read_pair_info)


def per_position_n_content(metrics, ranges, read_pair_info=None) -> dict:
    """report_modules.py:1202-1218"""
    ag = aggregate_count_matrix(list(metrics.base_count_table()), ranges, NUMBER_OF_NUCS)
    n = len(ag) // NUMBER_OF_NUCS
    out = [0.0] * n
    for i in range(n):
        t = ag[i * NUMBER_OF_NUCS:(i + 1) * NUMBER_OF_NUCS]
        if sum(t):
            out[i] = t[N] / sum(t)
    return dict(x_labels=stringify_ranges(ranges), n_content=out, read_pair_info=read_pair_info)


def per_sequence_gc_content(metrics, read_pair_info=None) -> dict:
    """report_modules.py:1304-1313"""
    gc = list(metrics.gc_content())
    smooth = [gc[2 * i] + gc[2 * i + 1] for i in range(50)] + [gc[100]]
    return dict(gc_content_counts=gc, smoothened_gc_content_counts=smooth, x_labels=[str(x) for x in range(101)],
                smoothened_x_labels=[str(x) for x in range(0, 101, 2)], read_pair_info=read_pair_info)


def per_sequence_quality_scores(metrics, read_pair_info=None) -> dict:
    """report_modules.py:1036-1038"""
    counts = list(metrics.phred_scores())
    return dict(average_quality_counts=counts, x_labels=[str(x) for x in range(len(counts))],
                read_pair_info=read_pair_info)


def adapter_content(adapter_counter, adapters, ranges, sample_length: int = 100, read_pair_info=None) -> dict:
    """AdapterContent.from_adapter_counter_adapters_and_ranges, report_modules.py:1431-1482"""
    def accumulate(counts):
        total, out = 0, []
        for c in counts:
            total += c
            out.append(total)
        return out

    by_sequence = {a.sequence: a for a in adapters}
    names = [a.name for a in adapters]
    total = adapter_counter.number_of_sequences
    all_, front, end = [], [], []
    for sequence, fwd, rev in adapter_counter.get_counts():
        fwd, rev = list(fwd), list(rev)
        end_counts = list(reversed(rev))
        ad = by_sequence[sequence]
        per_range = [sum(fwd[a:b]) for a, b in ranges]
        if ad.sequence_position == "end":
            acc = accumulate(per_range)
        else:
            acc = list(reversed(accumulate(reversed(per_range))))
        all_.append([c * 100 / total for c in acc])
        end.append([c * 100 / total for c in accumulate(end_counts[-sample_length:])])
        front.append([c * 100 / total for c in reversed(accumulate(reversed(fwd[:sample_length])))])
    return dict(x_labels=stringify_ranges(ranges), adapter_content=[list(x) for x in zip(names, all_)],
                front_adapter_content=[list(x) for x in zip(names, front)],
                end_adapter_content=[list(x) for x in zip(names, end)], read_pair_info=read_pair_info)


_DUP_SLICES = collections.OrderedDict([
    ("1", (1, 2)), ("2", (2, 3)), ("3", (3, 4)), ("4", (4, 5)), ("5", (5, 6)), ("6-10", (6, 11)),
    ("11-20", (11, 21)), ("21-30", (21, 31)), ("31-50", (31, 51)), ("51-100", (51, 101)),
    ("101-500", (101, 501)), ("501-1000", (501, 1001)), ("1001-5000", (1001, 5001)),
    ("5001-10000", (5001, 10_001)), ("10001-50000", (10_001, 50_001)), ("> 50000", (50_001, None))])


def duplication_fractions(dedup) -> dict:
    """DuplicationCounts.from_dedup_estimator, report_modules.py:1693-1756"""
    categories = collections.Counter(int(c) for c in dedup.duplication_counts())
    weights = [0] * 50002
    for duplication, count in categories.items():
        if duplication > 50_000:
            weights[50_001] += count * duplication
        else:
            weights[duplication] = count * duplication
    total = max(sum(weights), 1)
    fractions = {k: sum(weights[a:b]) / total for k, (a, b) in _DUP_SLICES.items()}
    total_sequences = sum(d * c for d, c in categories.items())
    return dict(tracked_unique_sequences=dedup.tracked_sequences, duplication_counts=[list(x) for x in sorted(categories.items())],
                remaining_fraction=sum(categories.values()) / max(total_sequences, 1),
                estimated_duplication_fractions=fractions,
                fingerprint_front_sequence_length=dedup.front_sequence_length,
                fingerprint_back_sequence_length=dedup.back_sequence_length,
                fingerprint_front_sequence_offset=dedup.front_sequence_offset,
                fingerprint_back_sequence_offset=dedup.back_sequence_offset)


_COMPLEMENT = str.maketrans("ACGTN", "TGCAN")


def overrepresented_sequences(seqdup, fraction_threshold=DEFAULT_FRACTION_THRESHOLD, min_threshold=DEFAULT_MIN_THRESHOLD,
                              max_threshold=DEFAULT_MAX_THRESHOLD, read_pair_info=None) -> dict:
    """OverRepresentedSequences.from_sequence_duplication, report_modules.py:1899-1928, without the
    identification of the sequences (most_matches, max_matches, best_match: None)"""
    rows = [dict(count=c, fraction=f, sequence=s, revcomp_sequence=s.translate(_COMPLEMENT)[::-1],
                 most_matches=None, max_matches=None, best_match=None)
            for c, f, s in seqdup.overrepresented_sequences(fraction_threshold, min_threshold, max_threshold)]
    return dict(overrepresented_sequences=rows, max_unique_fragments=seqdup.max_unique_fragments,
                sample_every=seqdup.sample_every, collected_fragments=seqdup.collected_unique_fragments,
                sequence_length=seqdup.fragment_length, total_fragments=seqdup.total_fragments,
                total_sequences=seqdup.number_of_sequences, sampled_sequences=seqdup.sampled_sequences,
                read_pair_info=read_pair_info)


def qc_modules(metrics, ranges, read_pair_info=None) -> Dict[str, dict]:
    suffix = "_read2" if read_pair_info == READ2 else ""
    base = list(metrics.base_count_table())
    return {
        "summary" + suffix: summary(metrics, ranges, read_pair_info),
        "sequence_length_distribution" + suffix: sequence_length_distribution(base, metrics.number_of_reads, ranges, read_pair_info),
        "per_sequence_quality_scores" + suffix: per_sequence_quality_scores(metrics, read_pair_info),
        "per_position_base_content" + suffix: per_position_base_content(metrics, ranges, read_pair_info),
        "per_position_n_content" + suffix: per_position_n_content(metrics, ranges, read_pair_info),
        "per_sequence_gc_content" + suffix: per_sequence_gc_content(metrics, read_pair_info),
    }


def report(modules: Dict[str, object], filename: str, filename_reverse: Optional[str] = None,
           graph_resolution: int = 200) -> Dict[str, dict]:
    """calculate_stats + report_modules_to_dict (report_modules.py:2607-2682, 2461-2471) over what
    driver.run returns"""
    def size(p):
        try:
            return os.path.getsize(p)
        except OSError:
            return 0

    m1 = modules["metrics"]
    info1 = READ1 if filename_reverse else None
    ranges = data_ranges_of(m1.max_length, graph_resolution)
    out: Dict[str, dict] = {"meta": dict(sequali_version="sequali_amd", filename=os.path.basename(filename),
                                         filesize=size(filename),
                                         filename_read2=os.path.basename(filename_reverse) if filename_reverse else None,
                                         filesize_read2=size(filename_reverse) if filename_reverse else None)}
    out.update(qc_modules(m1, ranges, info1))
    out["overrepresented_sequences"] = overrepresented_sequences(modules["sequence_duplication"], read_pair_info=info1)
    out["duplication_fractions"] = duplication_fractions(modules["dedup_estimator"])
    if modules.get("adapter_counter") is not None:
        out["adapter_content"] = adapter_content(modules["adapter_counter"], modules["adapters"], ranges, read_pair_info=info1)
    if modules.get("insert_size_metrics") is not None:
        out["insert_size_metrics"] = dict(insert_sizes=[int(x) for x in modules["insert_size_metrics"].insert_sizes()])
    m2 = modules.get("metrics_reverse")
    if m2 is not None and modules.get("sequence_duplication_reverse") is not None:
        out.update(qc_modules(m2, data_ranges_of(m2.max_length, graph_resolution), READ2))
        out["overrepresented_sequences_read2"] = overrepresented_sequences(modules["sequence_duplication_reverse"], read_pair_info=READ2)
    return out
