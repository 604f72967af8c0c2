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
