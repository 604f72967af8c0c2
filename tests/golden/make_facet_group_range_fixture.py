"""Generates tests/golden/facet_group_range.json from the REFERENCE's own fixtures + assertions (run in the build container):
 * grouped facet counts: documents = /root/reference/test/group_documents.jsonl, expected = CollectionGroupingTest.GroupingBasics
   (/root/reference/test/collection_grouping_test.cpp:71-110): q = *, facet `brand`, group_by `size` -> Beta 3, Omega 3, Xorp 2, Zeta 1
   (= the number of distinct sizes per brand; the plain document counts would be 3, 4, 2, 1);
 * range facets: CollectionFacetingTest.RangeFacetTest (/root/reference/test/collection_faceting_test.cpp:1500-1590: `visitors` with
   Busy:[0, 200000], VeryBusy:[200000, 500000]; "Karnataka" = documents 0, 1 -> Busy 1, VeryBusy 1; "Gujarat" = document 4 -> VeryBusy 1)
   and CollectionFacetingTest.RangeFacetTestWithGroupBy (:3419-3524: "Karnataka" -> VeryBusy 2; q = * grouped by `rating` -> VeryBusy 2, Busy 1).
 * float range facets: RangeFacetsFloatRange (:1839-1891), RangeFacetsMinMaxRange (:1893-1945: an open bound = INT64_MIN / INT64_MAX, src/collection.cpp:7547-7569)
   and RangeFacetRangeNegativeRanges (:1986-2043); values and bounds as Index::float_to_int64_t keys (src/index.cpp:266-274), like the sort index holds them.
Value hashes = crc32 of the value's string (any injective map works); the documents of the two range tests are inline in the tests."""
import json
import os
import zlib

h = lambda v: zlib.crc32(str(v).encode()) & 0xFFFFFFFF
docs = [json.loads(l) for l in open("/root/reference/test/group_documents.jsonl") if l.strip()]
out = {"source": "test/group_documents.jsonl + test/collection_grouping_test.cpp:71-110; test/collection_faceting_test.cpp:1500-1590, 3419-3524",
       "grouping_basics": {
           "brand_hashes": [[h(d["brand"])] if "brand" in d else [] for d in docs],
           "brand_values": {b: h(b) for b in sorted({d["brand"] for d in docs if "brand" in d})},
           "size_hashes": [[d["size"]] for d in docs],                   # the group_by field's facet hashes — an int32 field's hash IS its value (src/index.cpp:1733) —: distinct id = hash_combine(1, hash), the reference's own ids
           "result_ids": list(range(len(docs))),
           "expected_grouped": {"Beta": 3, "Omega": 3, "Xorp": 2, "Zeta": 1},
           # the grouped hits of the same request (:76-96; group_limit 2, default_sorting_field rating): found_docs 12, found 3, groups in this order
           "rating_keys": None,
           "expected_groups": [{"size": 11, "found": 2, "hits": [5, 1]}, {"size": 10, "found": 7, "hits": [4, 3]}, {"size": 12, "found": 3, "hits": [2, 8]}]},
       "range_facet_test": {
           "visitors": [235486, 187654, 174684, 246676, 345878],
           "ranges": [[200000, 0], [500000, 200000]],                     # (upper, lower): Busy, VeryBusy
           "karnataka_ids": [0, 1], "karnataka_expected": [1, 1],
           "gujarat_ids": [4], "gujarat_expected": [0, 1]},
       "range_facet_with_group_by": {
           "visitors": [235486, 201022, 174684, 246676, 345878],
           "rating_hashes": [[h(4.5)], [h(4.5)], [h(3.8)], [h(4.5)], [h(3.8)]],
           "ranges": [[200000, 0], [500000, 200000]],
           "karnataka_ids": [0, 1], "karnataka_expected": [0, 2],
           "all_ids": [0, 1, 2, 3, 4], "all_grouped_expected": [1, 2]}}
import struct
I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


def f2i(x):                                   # Index::float_to_int64_t (src/index.cpp:266-274)
    i = struct.unpack("<i", struct.pack("<f", x))[0]
    return i ^ 0x7FFFFFFF if i < 0 else i


inches = [f2i(v) for v in (32.4, 55, 55.6)]
nrr = [f2i(v) for v in (1.353, -0.193, -0.400, -0.969, -1.048, -1.248, -1.253, 1.481)]
out["float_ranges"] = [
    {"test": "RangeFacetsFloatRange small:[0, 55.5]", "vals": inches, "ranges": [[f2i(55.5), f2i(0.0)]], "expected": [2]},
    {"test": "RangeFacetsFloatRange big:[55, 55.6]", "vals": inches, "ranges": [[f2i(55.6), f2i(55.0)]], "expected": [1]},
    {"test": "RangeFacetsMinMaxRange small:[0, 55], large:[55, ]", "vals": inches, "ranges": [[f2i(55.0), f2i(0.0)], [I64_MAX, f2i(55.0)]], "expected": [1, 2]},
    {"test": "RangeFacetsMinMaxRange small:[,55]", "vals": inches, "ranges": [[f2i(55.0), I64_MIN]], "expected": [1]},
    {"test": "RangeFacetRangeNegativeRanges poor:[-1.5,-1], decent:[-1,0], good:[0,2]", "vals": nrr,
     "ranges": [[f2i(-1.0), f2i(-1.5)], [f2i(0.0), f2i(-1.0)], [f2i(2.0), f2i(0.0)]], "expected": [3, 3, 2]}]
out["grouping_basics"]["rating_keys"] = [f2i(d["rating"]) for d in docs]
out["grouping_basics"]["sizes"] = [d["size"] for d in docs]
# the second request of GroupingBasics (:112-148): group_by rating (a float field's facet hash = the float's bits, src/index.cpp:791), sort_by size DESC, group_limit 2:
# found_docs 12, found 7, seven groups; the test asserts groups 0, 1, 5 and 6
out["grouping_basics"]["rating_hashes"] = [[struct.unpack("<I", struct.pack("<f", d["rating"]))[0]] for d in docs]
out["grouping_basics"]["by_rating_expected"] = {"n_groups": 7, "groups": {"0": {"found": 1, "hits": [8]}, "1": {"found": 4, "hits": [6, 1]},
                                                                       "5": {"found": 1, "hits": [9]}, "6": {"found": 1, "hits": [0]}}}
# CollectionGroupingTest.GroupingCompoundKey (:150-215): group_by size + brand (brand is optional: documents 10 and 11 have none; group_missing_values = true keeps them in the
# group of their size alone), group_limit 2, default sort (rating desc): found_docs 12, found 10; groups 0, 1, 2 and 5 as asserted; facet counts of `brand` = groups per brand.
# A string field's facet ids are handed out in order of first appearance (++next_facet_id, src/facet_index.cpp:38-41): Omega 1, Beta 2, Xorp 3, Zeta 4.
brand_ids = {}
for d in docs:
    if "brand" in d and d["brand"] not in brand_ids:
        brand_ids[d["brand"]] = len(brand_ids) + 1
out["compound_key"] = {"brand_ids": brand_ids, "brand_hashes": [[brand_ids[d["brand"]]] if "brand" in d else [] for d in docs],
                       "n_groups": 10, "groups": {"0": {"found": 1, "hits": [5]}, "1": {"found": 1, "hits": [4]}, "2": {"found": 2, "hits": [3, 0]}, "5": {"found": 2, "hits": [10, 11]}},
                       "expected_grouped_facets": {"Beta": 3, "Omega": 3, "Xorp": 2, "Zeta": 1}}
# CollectionGroupingTest.GroupingWithGropLimitOfOne (:372-411): group_by brand (optional; the two unbranded documents form ONE group under the default group_missing_values = true),
# group_limit 1: found_docs 12, found 5; every brand's facet count is 1 (one group each).
out["group_limit_of_one"] = {"n_groups": 5, "groups": [{"found": 3, "hits": [5]}, {"found": 4, "hits": [3]}, {"found": 2, "hits": [8]}, {"found": 2, "hits": [10]}, {"found": 1, "hits": [9]}],
                             "expected_grouped_facets": {"Beta": 1, "Omega": 1, "Xorp": 1, "Zeta": 1}}
# CollectionGroupingTest.ControlMissingValues (:646-715): four documents, brand = Omega, null, null, Omega; no sort field (order = seq_id desc); group_limit 2.
# group_missing_values = false: three groups — Omega (3, 0), then documents 2 and 1 on their own; true (the default): two groups — Omega (3, 0) and the missing ones (2, 1)
out["control_missing_values"] = {"brand_hashes": [[1], [], [], [1]],
                                 "gmv_false": [{"hits": [3, 0]}, {"hits": [2]}, {"hits": [1]}],
                                 "gmv_true": [{"hits": [3, 0]}, {"hits": [2, 1]}]}
# CollectionFacetingTest.FacetStatOnFloatFields (test/collection_faceting_test.cpp:645-712) on test/float_documents.jsonl: stats of `average` (float: the facet hash is the
# float's bits) over all 7 documents: min -21.3799991607666, max 300, sum 277.8160007725237, avg 39.68800011036053 (ASSERT_FLOAT_EQ)
fdocs = [json.loads(l) for l in open("/root/reference/test/float_documents.jsonl") if l.strip()]
out["float_stats"] = {"average_bits": [[struct.unpack("<I", struct.pack("<f", d["average"]))[0]] for d in fdocs],
                      "expected": {"min": -21.3799991607666, "max": 300.0, "sum": 277.8160007725237, "avg": 39.68800011036053, "count": 7}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "facet_group_range.json"), "w"), indent=1)
print(len(docs), "documents")
