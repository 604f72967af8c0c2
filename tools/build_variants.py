"""Experiment tooling: builds libtsgpu variants with -D knobs applied to tsgpu.hip (the keyword kernels) in parallel.
usage: python tools/build_variants.py [--tu tsgpu_vec.hip] name1:-DX=1,-DY=2 name2:-DZ=3 ...  ->  typesense_amd/variants/libtsgpu_<name>.so"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from typesense_amd import build as Bd

TU = None

def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(ROOT, "typesense_amd", "variants", "libtsgpu_%s.so" % name)
    Bd.build(force=False, extra_flags=tuple(f for f in flags.split(",") if f), out=out, only_kw=True, only=TU)
    return out

if __name__ == "__main__":
    Bd.build()                                            # default objects first (the variants reuse the other translation units)
    args = sys.argv[1:]
    if args and args[0] == "--tu":
        TU = args[1]; args = args[2:]
    os.makedirs(os.path.join(ROOT, "typesense_amd", "variants"), exist_ok=True)
    with ThreadPoolExecutor(8) as ex:
        for o in ex.map(one, args):
            print(o)
