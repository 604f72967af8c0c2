"""Regenerates the golden fixtures under tests/golden/ from the reference's own unit tests.

Runs only in the build container (needs /root/reference); the outputs are committed so that the GPU
box — where /root/reference does not exist — can use them.

  or_iterator_4sublists.txt : the six id lists of OrIteratorTest.IntersectTwoListsWith4SubLists
                              (/root/reference/test/or_iterator_test.cpp:84-160), expected AND = {3199, 6414, 13357}
  record_values.txt         : the (key,score) rows of TopsterTest.StableSorting
                              (/root/reference/test/resources/record_values.txt, used at test/topster_test.cpp:60-136)
"""
import re, shutil, pathlib

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent

src = (REF / "test/or_iterator_test.cpp").read_text()
start = src.index("TEST(OrIteratorTest, IntersectTwoListsWith4SubLists)")
end = src.index("TEST(OrIteratorTest, IntersectAndFilterThreeIts)")
body = src[start:end]
lists = []
for blk in re.findall(r"=\s*\{\s*((?:\{[^{}]*\}\s*,?\s*)+)\}\s*;", body):
    for inner in re.findall(r"\{([^{}]*)\}", blk):
        nums = [int(x) for x in re.findall(r"\d+", inner)]
        lists.append(nums)
lists = [l for l in lists if l != [0, 1, 3]]          # drop the shared `offsets` literal
assert len(lists) == 6, len(lists)
with open(OUT / "or_iterator_4sublists.txt", "w") as f:
    f.write("# from /root/reference/test/or_iterator_test.cpp:84-160; lines 1-3 = token 1 sublists, 4-6 = token 2 sublists\n")
    for l in lists:
        f.write(" ".join(map(str, l)) + "\n")

shutil.copyfile(REF / "test/resources/record_values.txt", OUT / "record_values.txt")
print("ok", [len(l) for l in lists])
