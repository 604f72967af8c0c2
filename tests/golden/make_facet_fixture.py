"""Generates tests/golden/facet_counts_tags.json from the REFERENCE's own fixture + assertions (run in the build container):
documents = /root/reference/test/numeric_array_documents.jsonl (`tags`: string[] facet), expected counts = the assertions of
CollectionFacetingTest.FacetCounts (/root/reference/test/collection_faceting_test.cpp:66-87): query "Jeremy" matches all 5
documents; tags -> gold 3, silver 3, bronze 2, FINE PLATINUM 1. Value hashes = crc32 of the value (any injective map works)."""
import json
import os
import zlib

REF = "/root/reference/test/numeric_array_documents.jsonl"
docs = [json.loads(l) for l in open(REF) if l.strip()]
out = {"source": "test/numeric_array_documents.jsonl + test/collection_faceting_test.cpp:66-87",
       "docs": [[zlib.crc32(t.encode()) & 0xFFFFFFFF for t in d["tags"]] for d in docs],
       "values": {t: zlib.crc32(t.encode()) & 0xFFFFFFFF for d in docs for t in d["tags"]},
       "result_ids": list(range(len(docs))),
       "expected_counts": {"gold": 3, "silver": 3, "bronze": 2, "FINE PLATINUM": 1}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "facet_counts_tags.json"), "w"), indent=1)
print(out["expected_counts"], len(docs))

# ---- second fixture: numeric facet stats + the value-index branch, same documents ----
# stats: CollectionFacetingTest.FacetCounts, test/collection_faceting_test.cpp:240-296 — `rating` (float): avg 4.880199885368347, min 0.0,
# max 9.99899959564209, sum 24.400999426841736 over 5 values; `timestamps` (int64[]): avg 1106321222, min 348974822, max 1453426022,
# sum 13275854664 (=> 12 values). Float "hashes" are the float's bits, int64 values go through an fhash -> int64 map (crc32 of the decimal string).
# value index: CollectionOptimizedFacetingTest.FacetCounts (test/collection_optimized_faceting_test.cpp:60-110): the same `tags` counts through
# facet_index_t::intersect; visiting order = counter_list (by total count; gold and silver tie at 3: the test pins gold first).
import struct
ratings = [struct.unpack("<I", struct.pack("<f", d["rating"]))[0] for d in docs]
ts_hash = lambda v: zlib.crc32(str(v).encode()) & 0xFFFFFFFF
ts_values = sorted({v for d in docs for v in d["timestamps"]})
tags_order = ["gold", "silver", "bronze", "FINE PLATINUM"]
out2 = {"source": "test/numeric_array_documents.jsonl + test/collection_faceting_test.cpp:240-296 + test/collection_optimized_faceting_test.cpp:60-110",
        "rating_bits": [[b] for b in ratings],
        "rating_expected": {"min": 0.0, "max": 9.99899959564209, "sum": 24.400999426841736, "avg": 4.880199885368347, "count": 5},
        "timestamps_hashes": [[ts_hash(v) for v in d["timestamps"]] for d in docs],
        "timestamps_map": sorted([[ts_hash(v), v] for v in ts_values]),
        "timestamps_expected": {"min": 348974822, "max": 1453426022, "sum": 13275854664, "avg": 1106321222, "count": 12},
        "tags_values": tags_order,
        "tags_value_ids": [[i for i, d in enumerate(docs) if t in d["tags"]] for t in tags_order],
        "tags_expected": [["gold", 3], ["silver", 3], ["bronze", 2], ["FINE PLATINUM", 1]],
        "result_ids": list(range(len(docs)))}
assert len({h for h, _ in out2["timestamps_map"]}) == len(ts_values)
json.dump(out2, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "facet_stats_values.json"), "w"), indent=1)
print(out2["rating_expected"], out2["timestamps_expected"])
