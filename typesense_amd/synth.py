"""Synthetic collections of SURVEY.md §8(d) / BASELINE.md: Zipf text corpus -> decoded posting lists in the
reference's encoding (src/index.cpp:1323-1348), `points` sort column, query sets. Data plumbing for bench.py and
the tests only (torch is used as a fast sort/scan engine on whatever device is available); no scoring here.
"""
import math
import numpy as np


def _torch():
    import torch
    return torch


def points_column(n_docs):
    """points = hash(seq_id) mod 1000, int64"""
    ids = np.arange(n_docs, dtype=np.uint64)
    return ((ids * np.uint64(2654435761)) % np.uint64(1000)).astype(np.int64)


def zipf_corpus_csr(n_docs, vocab, tokens_per_doc, seed, s=1.0, device=None, doc_base=0, doc_range=None):
    """Returns the CSR posting arrays for tsgpu_terms_load_csr / oracle load_posting (all numpy, host):
        term_ids[u32 n_terms] (rank, 1-based, only terms that occur), ids_ptr[u64 n_terms+1], ids[u32],
        offset_index[u64 n_postings] (absolute), off_ptr[u64 n_terms+1], offsets[u32]
    Document d = tokens_per_doc i.i.d. Zipf(s) draws over `vocab` terms at positions 0..tokens_per_doc-1.
    offsets of (term, doc) = position+1 ascending, then 0 when the term is the doc's last token.
    doc_range = (lo, hi): the postings of documents lo <= d < hi of the SAME collection (the whole collection is drawn with the
    same random stream, then cut): a doc-range shard of exactly the corpus an unsharded run indexes; ids stay global."""
    torch = _torch()
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ranks = torch.arange(1, vocab + 1, dtype=torch.float64, device=dev)
    w = ranks.pow(-s)
    cdf = torch.cumsum(w / w.sum(), 0)
    n_tok = n_docs * tokens_per_doc
    T = tokens_per_doc
    # sampled in slabs to bound temporary memory; key = term * n_tok + doc * T + pos
    keys = torch.empty(n_tok, dtype=torch.int64, device=dev)
    slab = 1 << 26
    for a in range(0, n_tok, slab):
        b = min(n_tok, a + slab)
        u = torch.rand(b - a, generator=g, device=dev, dtype=torch.float64)
        term = torch.searchsorted(cdf, u).clamp_(max=vocab - 1)
        keys[a:b] = term * n_tok + torch.arange(a, b, device=dev, dtype=torch.int64)
        del u, term
    keys, _ = torch.sort(keys)
    term = torch.div(keys, n_tok, rounding_mode="floor")
    rem = keys - term * n_tok
    del keys
    doc = torch.div(rem, T, rounding_mode="floor")
    pos = rem - doc * T
    del rem
    if doc_range is not None:
        sel = (doc >= int(doc_range[0])) & (doc < int(doc_range[1]))
        term, doc, pos = term[sel], doc[sel], pos[sel]
        del sel
    n = term.numel()
    # run = one (term, doc) posting
    new_run = torch.ones(n, dtype=torch.bool, device=dev)
    new_run[1:] = (term[1:] != term[:-1]) | (doc[1:] != doc[:-1])
    run_end = torch.ones(n, dtype=torch.bool, device=dev)
    run_end[:-1] = new_run[1:]
    zflag = run_end & (pos == T - 1)                       # last token of the doc -> trailing 0
    zcum = torch.cumsum(zflag.to(torch.int64), 0)
    out_pos = torch.arange(n, device=dev, dtype=torch.int64) + (zcum - zflag.to(torch.int64))   # exclusive scan
    n_off = int(n + (zcum[-1].item() if n else 0))
    offsets = torch.zeros(n_off, dtype=torch.int32, device=dev)
    offsets[out_pos] = (pos + 1).to(torch.int32)
    ids = (doc[new_run] + doc_base).to(torch.int32)
    offset_index = out_pos[new_run]
    run_term = term[new_run]
    counts = torch.bincount(run_term, minlength=vocab)
    ids_ptr = torch.zeros(vocab + 1, dtype=torch.int64, device=dev)
    ids_ptr[1:] = torch.cumsum(counts, 0)
    # offsets of term t end where term t+1's first posting starts
    n_post = ids.numel()
    off_ptr = torch.full((vocab + 1,), n_off, dtype=torch.int64, device=dev)
    has = counts > 0
    first_post = ids_ptr[:-1][has]
    off_ptr[:-1][has] = offset_index[first_post]
    # empty terms: inherit the next non-empty start (scan from the right)
    op = off_ptr.cpu().numpy().astype(np.uint64)
    cnt = counts.cpu().numpy()
    nxt = n_off
    for t in range(vocab - 1, -1, -1):
        if cnt[t] > 0:
            nxt = op[t]
        else:
            op[t] = nxt
    term_ids = np.arange(1, vocab + 1, dtype=np.uint32)
    return dict(term_ids=term_ids, ids_ptr=ids_ptr.cpu().numpy().astype(np.uint64),
                ids=ids.cpu().numpy().astype(np.uint32), offset_index=offset_index.cpu().numpy().astype(np.uint64),
                off_ptr=op, offsets=offsets.cpu().numpy().astype(np.uint32), n_docs=n_docs, n_postings=n_post)


def csr_term(csr, term_id):
    """decoded (ids, offset_index relative, offsets) of one term of a zipf_corpus_csr result"""
    t = int(term_id) - 1
    a, b = int(csr["ids_ptr"][t]), int(csr["ids_ptr"][t + 1])
    o0, o1 = int(csr["off_ptr"][t]), int(csr["off_ptr"][t + 1])
    return csr["ids"][a:b], (csr["offset_index"][a:b] - np.uint64(o0)).astype(np.uint32), csr["offsets"][o0:o1]


def keyword_queries(n_queries, n_tokens, rank_lo, rank_hi, seed):
    """distinct term ranks per query, log-uniform in [rank_lo, rank_hi]"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_queries, n_tokens), np.uint32)
    for i in range(n_queries):
        while True:
            r = np.exp(rng.uniform(math.log(rank_lo), math.log(rank_hi + 1), size=n_tokens)).astype(np.int64)
            r = np.clip(r, rank_lo, rank_hi)
            if len(set(r.tolist())) == n_tokens:
                out[i] = r
                break
    return out


def random_vectors(n, dim, seed, device=None, normalize=False):
    """i.i.d. N(0,1) fp32 rows, generated where they will live (torch tensor on `device`)"""
    torch = _torch()
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.empty((n, dim), dtype=torch.float32, device=dev)
    slab = max(1, (1 << 28) // dim)
    for a in range(0, n, slab):
        b = min(n, a + slab)
        x[a:b] = torch.randn((b - a, dim), generator=g, device=dev, dtype=torch.float32)
    if normalize:
        x /= (x.norm(dim=1, keepdim=True) + 1e-30)
    return x


def latent_vectors(n, dim, seed, latent=32, noise=0.3, device=None, seed_w=9):
    """unit rows with neighbourhood structure: x = normalise(W z + noise * e), z ~ N(0, I_latent), W a fixed dim x latent map, e ~ N(0, I_dim).
    (i.i.d. N(0,1) rows in 768 dimensions are all but equidistant — no graph index can have recall on them; embedding collections have a
    low intrinsic dimension, which this imitates. Used by the HNSW bench leg.)"""
    torch = _torch()
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gw = torch.Generator(device=dev); gw.manual_seed(seed_w)
    W = torch.randn((latent, dim), generator=gw, device=dev, dtype=torch.float32) / (latent ** 0.5)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    x = torch.empty((n, dim), dtype=torch.float32, device=dev)
    slab = max(1, (1 << 27) // dim)
    for a in range(0, n, slab):
        b = min(n, a + slab)
        z = torch.randn((b - a, latent), generator=g, device=dev, dtype=torch.float32)
        x[a:b] = z @ W + noise * torch.randn((b - a, dim), generator=g, device=dev, dtype=torch.float32)
    x /= (x.norm(dim=1, keepdim=True) + 1e-30)
    return x
