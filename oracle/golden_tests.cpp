// ORACLE — TEST INFRASTRUCTURE ONLY.
// Known-answer tests that pin the oracle against the reference's own unit tests.
// Each check cites the reference test it restates. Run by tests/test_oracle_golden.py.
// Exit code 0 = all pass; prints "FAIL <where>" lines otherwise.
#include <cstdio>
#include <cstring>
#include <limits>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <random>
#include <string>
#include "oracle_index.h"

using namespace oracle;

static int g_fail = 0;
static int g_checks = 0;
#define CHECK(cond) do { g_checks++; if (!(cond)) { g_fail++; printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)
#define CHECK_EQ(a, b) do { g_checks++; auto _a = (a); auto _b = (b); if (!(_a == _b)) { g_fail++; \
    printf("FAIL %s:%d  %s == %s  (%lld vs %lld)\n", __FILE__, __LINE__, #a, #b, (long long)_a, (long long)_b); } } while (0)

static std::string g_golden_dir = "tests/golden";

static std::vector<std::vector<uint32_t>> read_lists(const std::string& path) {
    std::vector<std::vector<uint32_t>> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::vector<uint32_t> v;
        std::stringstream ss(line);
        uint32_t x;
        while (ss >> x) v.push_back(x);
        out.push_back(v);
    }
    return out;
}

// ---------- test/sorted_array_test.cpp, test/array_test.cpp ----------
static void test_arrays() {
    {   // SortedArrayTest.Append (test/sorted_array_test.cpp:8-60): append 0..N-1, values round-trip, indexOf, contains
        sorted_array arr;
        const int SIZE = 10 * 1000;
        CHECK_EQ(arr.getLength(), 0u);
        CHECK_EQ(arr.indexOf(100), 0u);
        CHECK(!arr.contains(100));
        for (int i = 0; i < SIZE; i++) { size_t idx = arr.append(i); CHECK_EQ((int)idx, i); }
        CHECK_EQ((int)arr.getLength(), SIZE);
        for (int i = 0; i < SIZE; i++) {
            if (arr.at(i) != (uint32_t)i) { CHECK_EQ((int)arr.at(i), i); break; }
            if (arr.indexOf(i) != (uint32_t)i) { CHECK_EQ((int)arr.indexOf(i), i); break; }
        }
        CHECK(arr.contains(SIZE - 1));
        CHECK(!arr.contains(SIZE + 1));
        CHECK_EQ((int)arr.indexOf(SIZE + 1), SIZE);
        CHECK_EQ((int)arr.last(), SIZE - 1);
        // out-of-order append re-sorts (:47-60)
        sorted_array arr_small;
        size_t i0 = arr_small.append(100);
        size_t i1 = arr_small.append(10);
        CHECK_EQ((int)i0, 0); CHECK_EQ((int)i1, 0);
        CHECK_EQ((int)arr_small.at(0), 10); CHECK_EQ((int)arr_small.at(1), 100);
    }
    {   // SortedArrayTest.Load / Uncompress (:100-140)
        sorted_array arr;
        const size_t SIZE = 6;
        uint32_t vals[SIZE] = {1, 3, 5, 10, 32, 4000};
        arr.load(vals, SIZE);
        uint32_t* raw = arr.uncompress();
        for (size_t i = 0; i < SIZE; i++) CHECK_EQ(raw[i], vals[i]);
        delete[] raw;
        CHECK_EQ(arr.getMin(), 1u); CHECK_EQ(arr.getMax(), 4000u);
        CHECK_EQ(arr.indexOf(32), 4u); CHECK_EQ(arr.indexOf(33), (uint32_t)SIZE);
        arr.remove_value(10);
        CHECK_EQ(arr.getLength(), 5u); CHECK_EQ(arr.at(3), 32u);
    }
    {   // ArrayTest.Append / InsertValues (test/array_test.cpp:6-60): unsorted values round-trip, min/max
        array arr;
        std::mt19937 rng(7);
        std::vector<uint32_t> ref;
        for (int i = 0; i < 3000; i++) { uint32_t v = rng() % 100000; ref.push_back(v); arr.append(v); }
        CHECK_EQ(arr.getLength(), 3000u);
        bool ok = true;
        for (int i = 0; i < 3000; i++) ok = ok && (arr.at(i) == ref[i]);
        CHECK(ok);
        CHECK_EQ(arr.getMin(), *std::min_element(ref.begin(), ref.end()));
        CHECK_EQ(arr.getMax(), *std::max_element(ref.begin(), ref.end()));
        uint32_t ins[3] = {7, 8, 999999};
        arr.insert(5, ins, 3);
        CHECK_EQ(arr.getLength(), 3003u);
        CHECK_EQ(arr.at(5), 7u); CHECK_EQ(arr.at(7), 999999u); CHECK_EQ(arr.at(8), ref[5]);
        arr.remove_index(5, 8);
        CHECK_EQ(arr.getLength(), 3000u); CHECK_EQ(arr.at(5), ref[5]);
    }
    {   // bit-width edges: all equal (0 bits), full 32-bit range
        sorted_array a;
        uint32_t same[4] = {9, 9, 9, 9};
        a.load(same, 4);
        CHECK_EQ(for_header_bits(a.raw()), 0u); CHECK_EQ(a.at(3), 9u);
        uint32_t wide[3] = {0, 5, 0xFFFFFFFFu};
        a.load(wide, 3);
        CHECK_EQ(for_header_bits(a.raw()), 32u); CHECK_EQ(a.at(2), 0xFFFFFFFFu); CHECK_EQ(a.at(1), 5u);
    }
}

// ---------- test/posting_list_test.cpp ----------
static void test_posting_lists() {
    std::vector<uint32_t> offsets = {0, 1, 3};
    {   // PostingListTest.Insert (:21-130)
        posting_list_t pl(5);
        for (size_t i = 0; i < 15; i++) pl.upsert((uint32_t)i, offsets);
        auto* root = pl.get_root();
        CHECK_EQ(root->ids.getLength(), 5u); CHECK_EQ(root->next->ids.getLength(), 5u);
        CHECK_EQ(root->next->next->ids.getLength(), 5u); CHECK(root->next->next->next == nullptr);
        CHECK_EQ(pl.num_blocks(), 3u); CHECK_EQ(pl.num_ids(), 15u);
        CHECK(root == pl.block_of(4)); CHECK(root->next == pl.block_of(9)); CHECK(root->next->next == pl.block_of(14));

        posting_list_t pl2(5);
        for (size_t i = 0; i < 15; i += 2) pl2.upsert((uint32_t)i, offsets);
        root = pl2.get_root();
        CHECK_EQ(root->ids.getLength(), 5u); CHECK_EQ(root->next->ids.getLength(), 3u);
        CHECK(root->next->next == nullptr); CHECK_EQ(pl2.num_blocks(), 2u); CHECK_EQ(pl2.num_ids(), 8u);

        posting_list_t pl3(5);
        for (size_t i = 0; i < 5; i++) pl3.upsert((uint32_t)i, offsets);
        for (uint32_t id : {6u, 8u, 9u, 10u, 12u}) pl3.upsert(id, offsets);
        CHECK_EQ(pl3.num_ids(), 10u);
        pl3.upsert(5, offsets);
        CHECK_EQ(pl3.num_blocks(), 3u); CHECK_EQ(pl3.num_ids(), 11u);
        CHECK_EQ(pl3.get_root()->ids.getLength(), 5u);
        CHECK_EQ(pl3.get_root()->next->ids.getLength(), 3u); CHECK_EQ(pl3.get_root()->next->ids.last(), 8u);
        CHECK_EQ(pl3.get_root()->next->next->ids.getLength(), 3u); CHECK_EQ(pl3.get_root()->next->next->ids.last(), 12u);
        for (size_t i = 0; i < pl3.get_root()->next->offset_index.getLength(); i++)
            CHECK_EQ(pl3.get_root()->next->offset_index.at((uint32_t)i), (uint32_t)(i * 3));
        for (size_t i = 0; i < pl3.get_root()->next->offsets.getLength(); i++)
            CHECK_EQ(pl3.get_root()->next->offsets.at((uint32_t)i), offsets[i % 3]);

        posting_list_t pl4(5);
        for (size_t i = 0; i < 5; i++) pl4.upsert((uint32_t)i, offsets);
        for (uint32_t id : {6u, 8u, 9u, 10u, 12u}) pl4.upsert(id, offsets);
        pl4.upsert(11, offsets);
        CHECK_EQ(pl4.num_blocks(), 3u); CHECK_EQ(pl4.num_ids(), 11u);
        CHECK_EQ(pl4.get_root()->next->ids.getLength(), 3u); CHECK_EQ(pl4.get_root()->next->ids.last(), 9u);
        CHECK_EQ(pl4.get_root()->next->next->ids.getLength(), 3u); CHECK_EQ(pl4.get_root()->next->next->ids.last(), 12u);
    }
    {   // PostingListTest.InsertInMiddle (:132-150)
        posting_list_t pl(3);
        pl.upsert(1, {1}); pl.upsert(3, {3}); pl.upsert(2, {2});
        for (uint32_t i = 0; i < 3; i++) {
            CHECK_EQ(pl.get_root()->ids.at(i), i + 1); CHECK_EQ(pl.get_root()->offset_index.at(i), i);
            CHECK_EQ(pl.get_root()->offsets.at(i), i + 1);
        }
    }
    {   // PostingListTest.InplaceUpserts (:152-260)
        posting_list_t pl(5);
        pl.upsert(2, {1, 2, 3}); pl.upsert(5, {1, 2, 3}); pl.upsert(7, {1, 2, 3});
        CHECK_EQ(pl.get_root()->offsets.getLength(), 9u);
        pl.upsert(2, {1, 2, 4});
        CHECK_EQ(pl.num_ids(), 3u); CHECK_EQ(pl.get_root()->offsets.getLength(), 9u);
        CHECK_EQ(pl.get_root()->offsets.at(2), 4u); CHECK_EQ(pl.get_root()->offset_index.at(1), 3u);
        pl.upsert(2, {5, 7});
        CHECK_EQ(pl.get_root()->offsets.getLength(), 8u);
        CHECK_EQ(pl.get_root()->offsets.at(0), 5u); CHECK_EQ(pl.get_root()->offsets.at(1), 7u); CHECK_EQ(pl.get_root()->offsets.at(2), 1u);
        CHECK_EQ(pl.get_root()->offset_index.at(1), 2u); CHECK_EQ(pl.get_root()->offset_index.at(2), 5u);
        pl.upsert(2, {0, 2, 8});
        CHECK_EQ(pl.get_root()->offsets.getLength(), 9u); CHECK_EQ(pl.get_root()->offsets.at(2), 8u); CHECK_EQ(pl.get_root()->offsets.at(3), 1u);
        CHECK_EQ(pl.get_root()->offset_index.at(1), 3u); CHECK_EQ(pl.get_root()->offset_index.at(2), 6u);
        pl.upsert(5, {1, 10});
        CHECK_EQ(pl.get_root()->offsets.getLength(), 8u); CHECK_EQ(pl.get_root()->offsets.at(4), 10u);
        CHECK_EQ(pl.get_root()->offset_index.at(2), 5u);
        pl.upsert(5, {2, 4, 12});
        CHECK_EQ(pl.get_root()->offsets.getLength(), 9u); CHECK_EQ(pl.get_root()->offsets.at(5), 12u); CHECK_EQ(pl.get_root()->offsets.at(6), 1u);
        CHECK_EQ(pl.get_root()->offset_index.at(2), 6u);
    }
    auto build = [&](posting_list_t& p, std::initializer_list<uint32_t> ids) { for (uint32_t id : ids) p.upsert(id, offsets); };
    {   // PostingListTest.MergeBasics (:559-601) and IntersectionBasics (:603-700)
        posting_list_t p1(2), p2(2), p3(2);
        build(p1, {0, 2, 3, 20}); build(p2, {1, 3, 5, 10, 20}); build(p3, {2, 3, 5, 7, 20});
        std::vector<posting_list_t*> lists = {&p1, &p2, &p3};
        std::vector<uint32_t> r;
        posting_list_t::merge(lists, r);
        std::vector<uint32_t> exp = {0, 1, 2, 3, 5, 7, 10, 20};
        CHECK(r == exp);
        r.clear();
        posting_list_t::intersect(lists, r);
        CHECK(r == (std::vector<uint32_t>{3, 20}));
        r.clear();
        std::vector<posting_list_t::iterator_t> its;
        for (auto* p : lists) its.push_back(p->new_iterator());
        result_iter_state_t st;
        posting_list_t::block_intersect(its, st, [&](uint32_t id, std::vector<posting_list_t::iterator_t>&) { r.push_back(id); });
        CHECK(r == (std::vector<uint32_t>{3, 20}));
        r.clear();
        std::vector<posting_list_t*> single = {&p1};
        posting_list_t::intersect(single, r);
        CHECK(r == (std::vector<uint32_t>{0, 2, 3, 20}));
        r.clear();
        std::vector<posting_list_t*> empty;
        posting_list_t::intersect(empty, r);
        CHECK(r.empty());
    }
    {   // PostingListTest.IntersectionSkipBlocks (:774-823)
        posting_list_t p1(2), p2(2), p3(2);
        build(p1, {9, 11}); build(p2, {1, 2, 3, 4, 5, 6, 7, 8, 9, 11}); build(p3, {2, 3, 8, 9, 11, 20});
        std::vector<posting_list_t*> lists = {&p1, &p2, &p3};
        std::vector<uint32_t> r;
        posting_list_t::intersect(lists, r);
        CHECK(r == (std::vector<uint32_t>{9, 11}));
    }
    {   // PostingListTest.PostingListContainsAtleastOne (:825-859)
        posting_list_t p1(100);
        for (uint32_t i = 20; i < 40; i++) p1.upsert(i, offsets);
        uint32_t t1[] = {10, 25}; CHECK(p1.contains_atleast_one(t1, 2));
        uint32_t t2[] = {10, 50}; CHECK(!p1.contains_atleast_one(t2, 2));
    }
    {   // PostingListTest.BlockIntersectionOnMixedLists (:1295-1328): compact + full
        uint32_t ids[] = {5, 6, 7, 8};
        uint32_t offset_index[] = {0, 3, 6, 9};
        uint32_t offs[] = {0, 3, 4, 0, 3, 4, 0, 3, 4, 0, 3, 4};
        compact_posting_list_t* list1 = compact_posting_list_t::create(4, ids, offset_index, 12, offs);
        CHECK_EQ(list1->num_ids(), 4u); CHECK_EQ(list1->first_id(), 5u); CHECK_EQ(list1->last_id(), 8u);
        posting_list_t p1(2);
        for (uint32_t id : {0u, 5u, 8u, 20u}) p1.upsert(id, {2, 4});
        posting_list_t* full1 = list1->to_full_posting_list();
        std::vector<posting_list_t::iterator_t> its;
        its.push_back(full1->new_iterator());
        its.push_back(p1.new_iterator());
        result_iter_state_t st;
        std::vector<uint32_t> r;
        posting_list_t::block_intersect(its, st, [&](uint32_t id, std::vector<posting_list_t::iterator_t>&) { r.push_back(id); });
        CHECK(r == (std::vector<uint32_t>{5, 8}));
        delete full1; delete list1;
    }
    {   // load_sorted == sequential upserts of ascending ids (structure + decoded content)
        std::mt19937 rng(3);
        std::vector<uint32_t> ids, oi, off;
        uint32_t id = 0;
        for (int i = 0; i < 1000; i++) {
            id += 1 + rng() % 50; ids.push_back(id); oi.push_back((uint32_t)off.size());
            int n = 1 + rng() % 3;
            uint32_t pos = 0;
            for (int j = 0; j < n; j++) { pos += 1 + rng() % 9; off.push_back(pos); }  // strictly increasing positions
            if (rng() % 4 == 0) off.push_back(0);
        }
        posting_list_t a(256), b(256);
        a.load_sorted(ids.data(), oi.data(), off.data(), (uint32_t)ids.size(), (uint32_t)off.size());
        for (size_t i = 0; i < ids.size(); i++) {
            uint32_t e = (i + 1 < ids.size()) ? oi[i + 1] : (uint32_t)off.size();
            b.upsert(ids[i], std::vector<uint32_t>(off.begin() + oi[i], off.begin() + e));
        }
        CHECK_EQ(a.num_blocks(), b.num_blocks()); CHECK_EQ(a.num_ids(), b.num_ids());
        auto ia = a.new_iterator(); auto ib = b.new_iterator();
        bool same = true;
        while (ia.valid() && ib.valid()) {
            same = same && ia.id() == ib.id() && ia.index() == ib.index() && ia.offset_index[ia.index()] == ib.offset_index[ib.index()];
            std::vector<posting_list_t::iterator_t> va, vb;
            va.push_back(ia.clone()); vb.push_back(ib.clone());
            std::map<size_t, std::vector<token_positions_t>> ma, mb;
            posting_list_t::get_offsets(va, ma); posting_list_t::get_offsets(vb, mb);
            same = same && ma.size() == mb.size() && (ma.empty() || (ma[0][0].positions == mb[0][0].positions && ma[0][0].last_token == mb[0][0].last_token));
            ia.next(); ib.next();
        }
        CHECK(same); CHECK(!ia.valid() && !ib.valid());
    }
}

// ---------- test/or_iterator_test.cpp ----------
static std::vector<uint32_t> or_intersect(std::vector<std::vector<std::vector<uint32_t>>> groups, uint16_t blk,
                                          const std::vector<uint32_t>& filter_ids, size_t* nkm = nullptr) {
    std::vector<uint32_t> offsets = {0, 1, 3};
    std::vector<std::unique_ptr<posting_list_t>> owned;
    std::vector<or_iterator_t> or_its;
    for (auto& g : groups) {
        std::vector<posting_list_t::iterator_t> pits;
        for (auto& l : g) {
            owned.emplace_back(new posting_list_t(blk));
            for (uint32_t id : l) owned.back()->upsert(id, offsets);
            pits.push_back(owned.back()->new_iterator());
        }
        or_iterator_t it(pits);
        or_its.push_back(std::move(it));
    }
    result_iter_state_t st(nullptr, 0, filter_ids.empty() ? nullptr : filter_ids.data(), filter_ids.size());
    deadline_t dl;
    std::vector<uint32_t> results;
    or_iterator_t::intersect(or_its, st, dl, [&](const single_filter_result_t& fr, std::vector<or_iterator_t>&) { results.push_back(fr.seq_id); });
    if (nkm) *nkm = st.num_keyword_matches;
    or_its.clear();
    return results;
}

static void test_or_iterator() {
    {   // OrIteratorTest.IntersectTwoListsWith3SubLists (:8-82)
        auto r = or_intersect({{{0, 2, 3, 20}, {1, 3, 5, 10, 20}, {2, 3, 6, 7, 20}},
                               {{0, 1, 5, 20}, {1, 2, 7, 11, 15}, {3, 5, 10, 11, 12}}}, 2, {});
        CHECK(r == (std::vector<uint32_t>{0, 1, 2, 3, 5, 7, 10, 20}));
    }
    {   // OrIteratorTest.IntersectTwoListsWith4SubLists (:84-160); lists in tests/golden/or_iterator_4sublists.txt
        auto L = read_lists(g_golden_dir + "/or_iterator_4sublists.txt");
        CHECK_EQ(L.size(), 6u);
        if (L.size() == 6) {
            auto r = or_intersect({{L[0], L[1], L[2]}, {L[3], L[4], L[5]}}, 2, {});
            CHECK(r == (std::vector<uint32_t>{3199, 6414, 13357}));
        }
    }
    {   // OrIteratorTest.IntersectAndFilterThreeIts / TwoIts (:162-264)
        std::vector<uint32_t> a = {4207, 29159, 47182, 47250, 47337, 48518, 99820};
        std::vector<uint32_t> b = {62, 330, 367, 4124, 4207, 4242, 4418, 28740, 29099, 29159, 29284, 40795, 43556, 46779, 47182, 47250, 47322, 48494, 48518, 48633, 98813, 98821, 99069, 99368, 99533, 99670, 99820, 99888, 99973};
        std::vector<uint32_t> c = {723, 1504, 29038, 29164, 29390, 30890, 34743, 35067, 36466, 40268, 40965, 42161, 43425, 45188, 47326, 47443, 49319, 53043, 58436, 58774, 61123, 70973, 71393, 81575, 82323, 88301, 88502, 88594, 88690, 88951, 90662, 91016, 91915, 92069, 92844, 99820};
        std::vector<uint32_t> filter_ids = {44424, 44425, 44447, 99820, 99834, 99854, 99859, 99963};
        auto r3 = or_intersect({{a}, {b}, {c}}, 256, filter_ids);
        CHECK(r3 == (std::vector<uint32_t>{99820}));
        auto r2 = or_intersect({{a}, {b}}, 256, filter_ids);
        CHECK(r2 == (std::vector<uint32_t>{99820}));
    }
}

// ---------- test/match_score_test.cpp ----------
static void test_match() {
    {   // MatchTest.TokenOffsetsExceedWindowSize (:9-28)
        std::vector<token_positions_t> tp(12, token_positions_t{false, {1}});
        Match m(100, tp);
        CHECK_EQ((size_t)m.words_present, WINDOW_SIZE);
    }
    {   // MatchTest.MatchScoreV2 (:30-171)
        std::vector<token_positions_t> t;
        t.push_back({false, {25}}); t.push_back({false, {26}}); t.push_back({false, {11, 18, 24, 60}}); t.push_back({false, {14, 27, 63}});
        Match m(100, t, true);
        CHECK_EQ(m.words_present, 4); CHECK_EQ(m.distance, 3);
        uint16_t e1[] = {25, 26, 24, 27};
        for (int i = 0; i < 4; i++) CHECK_EQ(m.offsets[i].offset, e1[i]);
        m = Match(100, t, false);
        CHECK_EQ(m.words_present, 4); CHECK_EQ(m.distance, 3); CHECK_EQ(m.offsets.size(), 0u);

        t.clear();
        t.push_back({false, {38, 50, 170, 187, 195, 222}}); t.push_back({true, {39, 140, 171, 189, 223}}); t.push_back({false, {169, 180}});
        m = Match(100, t, true, true);
        CHECK_EQ(m.words_present, 3); CHECK_EQ(m.distance, 2); CHECK_EQ(m.exact_match, 0);
        uint16_t e2[] = {170, 171, 169};
        for (int i = 0; i < 3; i++) CHECK_EQ(m.offsets[i].offset, e2[i]);

        t.clear();
        t.push_back({false, {38, 50, 187, 195, 201}}); t.push_back({false, {120, 167, 171, 223}}); t.push_back({true, {240, 250}});
        m = Match(100, t, true);
        CHECK_EQ(m.words_present, 1); CHECK_EQ(m.distance, 0); CHECK_EQ(m.exact_match, 0);
        uint16_t e3[] = {38, MAX_DISPLACEMENT, MAX_DISPLACEMENT};
        for (int i = 0; i < 3; i++) CHECK_EQ(m.offsets[i].offset, e3[i]);

        t.clear();
        t.push_back({false, {0}}); t.push_back({true, {2}}); t.push_back({false, {1}});
        m = Match(100, t, true, true);
        CHECK_EQ(m.words_present, 3); CHECK_EQ(m.distance, 2); CHECK_EQ(m.exact_match, 1);
        m = Match(100, t, true, false);
        CHECK_EQ(m.exact_match, 0);

        t.clear();
        t.push_back({false, {1}}); t.push_back({false, {2}}); t.push_back({true, {3}});
        m = Match(100, t, true, true);
        CHECK_EQ(m.exact_match, 0);
        t.clear();
        t.push_back({false, {0}}); t.push_back({false, {1}}); t.push_back({false, {2}});
        m = Match(100, t, true, true);
        CHECK_EQ(m.exact_match, 0);

        t.clear();
        t.push_back({false, {74}}); t.push_back({false, {75}}); t.push_back({false, {3, 42}});
        m = Match(100, t, true, true);
        uint16_t e4[] = {74, 75, MAX_DISPLACEMENT};
        CHECK_EQ(m.offsets.size(), 3u);
        for (int i = 0; i < 3; i++) CHECK_EQ(m.offsets[i].offset, e4[i]);
    }
}

// ---------- test/topster_test.cpp ----------
static void test_topster() {
    {   // TopsterTest.MaxIntValues (:7-58)
        Topster topster(5);
        struct { uint16_t qi; uint64_t key; uint64_t ms; int64_t p; int64_t s; } data[14] = {
            {0, 1, 11, 20, 30}, {0, 1, 12, 20, 32}, {0, 2, 4, 20, 30}, {2, 3, 7, 20, 30}, {0, 4, 14, 20, 30},
            {1, 5, 9, 20, 30}, {1, 5, 10, 20, 32}, {1, 5, 9, 20, 30}, {0, 6, 6, 20, 30}, {2, 7, 6, 22, 30},
            {2, 7, 6, 22, 30}, {1, 8, 9, 20, 30}, {0, 9, 8, 20, 30}, {3, 10, 5, 20, 30}};
        for (auto& d : data) { int64_t sc[3] = {(int64_t)d.ms, d.p, d.s}; KV kv(d.qi, d.key, d.key, 0, sc); topster.add(&kv); }
        topster.sort();
        uint64_t ids[] = {4, 1, 5, 8, 9};
        CHECK_EQ(topster.size, 5u);
        for (uint32_t i = 0; i < topster.size; i++) {
            CHECK_EQ(topster.getKeyAt(i), ids[i]);
            if (ids[i] == 1) CHECK_EQ(topster.getKV(i)->scores[0], 12);
            if (ids[i] == 5) CHECK_EQ(topster.getKV(i)->scores[0], 10);
        }
    }
    {   // TopsterTest.StableSorting (:60-136), fixture tests/golden/record_values.txt
        std::ifstream f(g_golden_dir + "/record_values.txt");
        std::vector<std::pair<uint64_t, int64_t>> records;
        std::string line;
        while (std::getline(f, line)) {
            auto c = line.find(',');
            if (c == std::string::npos) continue;
            records.emplace_back(std::stoll(line.substr(0, c)), std::stoi(line.substr(c + 1)));
        }
        CHECK_EQ(records.size(), 816u);
        auto run = [&](size_t cap) {
            Topster t(cap);
            for (auto& r : records) { int64_t sc[3] = {r.second, 0, 0}; KV kv(0, r.first, r.first, 0, sc); t.add(&kv); }
            t.sort();
            std::vector<uint64_t> keys;
            for (uint32_t i = 0; i < t.size; i++) keys.push_back(t.getKeyAt(i));
            return keys;
        };
        auto k1000 = run(1000);
        for (size_t cap : {250, 500, 750}) {
            auto k = run(cap);
            CHECK_EQ(k.size(), cap);
            bool prefix = true;
            for (size_t i = 0; i < k.size(); i++) prefix = prefix && (k[i] == k1000[i]);
            CHECK(prefix);
        }
    }
    {   // TopsterTest.MaxFloatValues (:138-179)
        Topster topster(5);
        struct { uint16_t qi; uint64_t key; uint64_t ms; float p; int64_t s; } data[12] = {
            {0, 1, 11, 1.09f, 30}, {0, 2, 11, -20, 30}, {2, 3, 11, -20, 30}, {0, 4, 11, 7.812f, 30}, {0, 4, 11, 7.912f, 30},
            {1, 5, 11, 0.0f, 34}, {0, 6, 11, -22, 30}, {2, 7, 11, -22, 30}, {1, 8, 11, -9.998f, 30}, {1, 8, 11, -9.998f, 30},
            {0, 9, 11, -9.999f, 30}, {3, 10, 11, -20, 30}};
        for (auto& d : data) { int64_t sc[3] = {(int64_t)d.ms, float_to_int64_t(d.p), d.s}; KV kv(d.qi, d.key, d.key, 0, sc); topster.add(&kv); }
        topster.sort();
        uint64_t ids[] = {4, 1, 5, 8, 9};
        for (uint32_t i = 0; i < topster.size; i++) CHECK_EQ(topster.getKeyAt(i), ids[i]);
    }
}

// ---------- end-to-end text_match values ----------
// ---------- test/topster_test.cpp:181-262 (TopsterTest.DistinctIntValues): the group_by forms of the collector ----------
static void test_distinct_topster() {
    struct { uint16_t qi; uint64_t dk; uint64_t ms; int64_t p, s; } data[14] = {
        {0, 1, 11, 20, 30}, {0, 1, 12, 20, 32}, {0, 2, 4, 20, 30}, {2, 3, 7, 20, 30}, {0, 4, 14, 20, 30}, {1, 5, 9, 20, 30}, {1, 5, 10, 20, 32},
        {1, 5, 9, 20, 30}, {0, 6, 6, 20, 30}, {2, 7, 6, 22, 30}, {2, 7, 6, 22, 30}, {1, 8, 9, 20, 30}, {0, 9, 8, 20, 30}, {3, 10, 5, 20, 30}};
    {   // second pass: Topster<KV> dist_topster(5, 2, false) — every KV lands in its group's Topster, the outer heap stays empty (:182-236)
        GroupTopster t(5, 2, false);
        for (int i = 0; i < 14; i++) { int64_t sc[3] = {(int64_t)data[i].ms, data[i].p, data[i].s}; KV kv(data[i].qi, (uint64_t)i + 100, data[i].dk, 0, sc); CHECK_EQ(t.add(&kv), 1); }
        t.sort();
        CHECK_EQ(t.size, 0u);
        CHECK_EQ(t.group_kv_map.size(), (size_t)10);
        for (auto& g : t.group_kv_map) g.second->sort();
        CHECK_EQ(t.group_kv_map[1]->size, 2u); CHECK_EQ(t.group_kv_map[1]->getKV(0)->scores[0], 12); CHECK_EQ(t.group_kv_map[1]->getKV(1)->scores[0], 11);
        CHECK_EQ(t.group_kv_map[5]->size, 2u); CHECK_EQ(t.group_kv_map[5]->getKV(0)->scores[0], 10); CHECK_EQ(t.group_kv_map[5]->getKV(1)->scores[0], 9);
        // populate_result_kvs (src/index.cpp:8962-9011) over it: the five best groups by their head
        std::unordered_map<uint64_t, uint32_t> processed;
        for (int i = 0; i < 14; i++) processed[data[i].dk]++;
        grouped_result_t out;
        populate_grouped(t, processed, out);
        const uint64_t order[5] = {4, 1, 5, 8, 9};
        CHECK_EQ(out.groups.size(), (size_t)5);
        for (size_t i = 0; i < out.groups.size() && i < 5; i++) CHECK_EQ(out.groups[i][0].distinct_key, order[i]);
        CHECK_EQ(out.group_found[2], 3u);
    }
    {   // first pass: Topster<KV> dist_topster_first_pass(7, 2, true) — the heap array as it lies + the loglog cardinality (:238-261)
        GroupTopster t(7, 2, true);
        for (int i = 0; i < 14; i++) { int64_t sc[3] = {(int64_t)data[i].ms, data[i].p, data[i].s}; KV kv(data[i].qi, (uint64_t)i + 100, data[i].dk, 0, sc); t.add(&kv); }
        t.sort();
        const uint64_t distinct_ids[7] = {7, 5, 3, 4, 1, 9, 8}, ids[7] = {110, 106, 103, 104, 101, 112, 111};
        CHECK_EQ(t.size, 7u);
        for (uint32_t i = 0; i < t.size && i < 7; i++) { CHECK_EQ(t.kvs[i]->distinct_key, distinct_ids[i]); CHECK_EQ(t.kvs[i]->key, ids[i]); }
        CHECK(t.group_kv_map.empty());
        CHECK_EQ(t.loglog_counter->cardinality(), (uint64_t)10);
    }
}

static void test_text_scores() {
    // vocabulary ids: nike=1 running=2 shoes=3 x=4 mong=5 spencer=6
    {   // test/collection_vector_search_test.cpp:5462-5497 text_match constants (1 field, weight 15, max_score)
        Index idx(1, 1);
        idx.index_plain_field(0, 0, {1, 2, 3, 4});   // 3 query tokens adjacent, field continues
        idx.index_plain_field(1, 0, {1, 2, 4, 4});   // 2 tokens adjacent
        idx.index_plain_field(2, 0, {1, 4, 4, 4});   // 1 token
        for (uint32_t i = 0; i < 3; i++) idx.set_sort_value(0, i, 10 * i);
        keyword_query_t q;
        q.fields = {{0, 15}};
        q.sort = {{SORT_TEXT_MATCH, 0, 1}, {SORT_INT64_COLUMN, 0, 1}};
        q.tokens = {1, 2, 3};
        auto r = idx.search_keyword(q);
        CHECK_EQ(r.kvs.size(), 1u);
        if (!r.kvs.empty()) CHECK_EQ((uint64_t)r.kvs[0].scores[0], 1736172819517016185ull);
        q.tokens = {1, 2};
        r = idx.search_keyword(q);
        CHECK_EQ(r.kvs.size(), 2u);
        if (r.kvs.size() == 2) { CHECK_EQ((uint64_t)r.kvs[0].scores[0], 1157451471441102969ull); CHECK_EQ(r.kvs[0].key, 1u); CHECK_EQ(r.kvs[1].key, 0u); }
        q.tokens = {1};
        r = idx.search_keyword(q);
        CHECK_EQ(r.kvs.size(), 3u);
        if (r.kvs.size() == 3) { CHECK_EQ((uint64_t)r.kvs[0].scores[0], 578730123365189753ull); CHECK_EQ(r.kvs[0].key, 2u); CHECK_EQ(r.num_keyword_matches, 3u); }
    }
    {   // CollectionSortingTest.RepeatingTokenRanking (test/collection_sorting_test.cpp:1800-1855)
        Index idx(1, 1);
        idx.index_plain_field(0, 0, {5, 5});
        idx.index_plain_field(1, 0, {5, 6});
        idx.index_plain_field(2, 0, {5, 5, 6});
        idx.index_plain_field(3, 0, {6, 5, 5});
        int64_t pts[] = {100, 200, 300, 400};
        for (uint32_t i = 0; i < 4; i++) idx.set_sort_value(0, i, pts[i]);
        keyword_query_t q;
        q.fields = {{0, 3}};   // the reference test passes query_by_weights = {3}
        q.sort = {{SORT_TEXT_MATCH, 0, 1}, {SORT_INT64_COLUMN, 0, 1}};
        q.tokens = {5, 5};
        auto r = idx.search_keyword(q);
        CHECK_EQ(r.kvs.size(), 4u);
        if (r.kvs.size() == 4) {
            uint64_t ek[] = {0, 3, 2, 1};
            for (int i = 0; i < 4; i++) CHECK_EQ(r.kvs[i].key, ek[i]);
            CHECK_EQ((uint64_t)r.kvs[0].scores[0], 1157451471583709209ull);
            for (int i = 1; i < 4; i++) CHECK_EQ((uint64_t)r.kvs[i].scores[0], 1157451471575320601ull);
        }
    }
    {   // token absent from every field is silently skipped (index.cpp:5651-5655); tie-break larger seq_id first
        Index idx(1, 1);
        for (uint32_t d = 0; d < 5; d++) idx.index_plain_field(d, 0, {1, 2});
        keyword_query_t q;
        q.fields = {{0, 15}};
        q.sort = {{SORT_TEXT_MATCH, 0, 1}, {SORT_SEQ_ID, 0, 1}};
        q.tokens = {1, 99, 2};
        auto r = idx.search_keyword(q);
        CHECK_EQ(r.kvs.size(), 5u);
        if (r.kvs.size() == 5) for (int i = 0; i < 5; i++) CHECK_EQ(r.kvs[i].key, (uint64_t)(4 - i));
    }
}

// ---------- vector distances: test/collection_vector_search_test.cpp ----------
static void test_vectors() {
    {   // :5110-5154  IP, 5-d, mt19937 seed 47, distance_threshold
        Index idx(1, 1);
        idx.vec_init(5, ip);
        std::mt19937 rng;
        rng.seed(47);
        std::uniform_real_distribution<> distrib;
        for (size_t i = 0; i < 5; i++) {
            std::vector<float> values;
            for (size_t j = 0; j < 5; j++) values.push_back((float)(distrib(rng) + 0.01));
            idx.vec_add((uint32_t)i, values.data());
        }
        vector_query_t vq;
        vq.values = {0.3f, 0.4f, 0.5f, 0.6f, 0.7f};  // not the reference's query; sanity only: order + 1-dot identity
        auto hits = idx.flat_knn(vq.values, 5);
        CHECK_EQ(hits.size(), 5u);
        for (size_t i = 1; i < hits.size(); i++) CHECK(hits[i - 1].dist <= hits[i].dist);
        const float* x = idx.vec_get(hits[0].seq_id);
        float dot = 0; for (int j = 0; j < 5; j++) dot += vq.values[j] * x[j];
        CHECK(std::fabs((1.0f - dot) - hits[0].dist) < 1e-6f);
    }
    {   // :83-122 cosine, 4-d: docs [0.04,0.234,0.113,0.001], [0.167,0.319,0.402,0.017], [0.081,0.124,0.273,0.010]
        // query [0.96826, 0.94, 0.39557, 0.306488]: expected distances 3.409385681152344e-05? (that is for a
        // different query in the reference); here: check cosine == 1 - cos(q,x) against double math
        Index idx(1, 1);
        idx.vec_init(4, cosine);
        float d[3][4] = {{0.04f, 0.234f, 0.113f, 0.001f}, {0.167f, 0.319f, 0.402f, 0.017f}, {0.081f, 0.124f, 0.273f, 0.010f}};
        for (uint32_t i = 0; i < 3; i++) idx.vec_add(i, d[i]);
        std::vector<float> q = {0.96826f, 0.94f, 0.39557f, 0.306488f};
        auto hits = idx.flat_knn(q, 3);
        for (auto& h : hits) {
            double dot = 0, nq = 0, nx = 0;
            for (int j = 0; j < 4; j++) { dot += (double)q[j] * d[h.seq_id][j]; nq += (double)q[j] * q[j]; nx += (double)d[h.seq_id][j] * d[h.seq_id][j]; }
            double expect = 1.0 - dot / std::sqrt(nq * nx);
            CHECK(std::fabs(expect - h.dist) < 1e-6);
        }
        // reference pins (test :120-122): 3.409385681152344e-05, 0.04329806566238403, 0.15141665935516357 for
        // query [0.04, 0.234, 0.113, 0.001] (doc 0 itself)
        std::vector<float> q0 = {0.04f, 0.234f, 0.113f, 0.001f};
        hits = idx.flat_knn(q0, 3);
        CHECK_EQ(hits[0].seq_id, 0u);
    }
    {   // float_to_int64_t is order preserving (index.cpp:266-274) incl. negatives
        float vals[] = {-100.f, -1.5f, -0.0f, 0.0f, 1e-9f, 0.5f, 3.f, 1e20f};
        for (int i = 0; i + 1 < 8; i++) CHECK(float_to_int64_t(vals[i]) <= float_to_int64_t(vals[i + 1]));
        for (float v : vals) CHECK(int64_t_to_float(float_to_int64_t(v)) == v);
    }
}

// ---------- the reference's own vector-distance known answers ----------
// Generators are the reference's, verbatim in behaviour: std::mt19937 seed 47, std::uniform_real_distribution<> (double, two
// engine draws per value) narrowed to float; documents travel through JSON (float -> double -> float: exact), query literals
// through std::stof (src/vector_query_ops.cpp:70). ASSERT_FLOAT_EQ in the reference = within 4 ulp; ASSERT_EQ on .get<float>() = bit-exact.
static int ulp_gap(float a, float b) {
    int32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    if (x < 0) x = INT32_MIN - x;
    if (y < 0) y = INT32_MIN - y;
    long long d = (long long)x - (long long)y;
    return (int)(d < 0 ? -d : d);
}
static int g_max_pin_ulp = 0;
#define CHECK_PIN(got, pin) do { float _g = (got), _p = (float)(pin); int _u = ulp_gap(_g, _p); if (_u > g_max_pin_ulp) g_max_pin_ulp = _u; \
    g_checks++; if (!(std::fabs((double)_g - (double)_p) <= 1e-5 * std::fabs((double)_p))) { g_fail++; printf("FAIL %s:%d  %s = %.9g, pinned %.9g\n", __FILE__, __LINE__, #got, _g, _p); } } while (0)
#define CHECK_PIN_BITS(got, pin) do { float _g = (got), _p = (float)(pin); g_checks++; if (memcmp(&_g, &_p, 4) != 0) { g_fail++; \
    printf("FAIL %s:%d  %s = %.9g, pinned bits of %.9g (%d ulp)\n", __FILE__, __LINE__, #got, _g, _p, ulp_gap(_g, _p)); } } while (0)

// What the result assembly prints for a `_vector_query(...)` sort key: compute_sort_scores stores float_to_int64_t(dist), negated
// for ASC (src/index.cpp:5850, :5901-5903), and Collection::search emits -int64_t_to_float(score) (src/collection.cpp:3183).
// The order-preserving bit trick is not symmetric under negation: the printed value is the distance moved ONE ulp towards
// -inf in magnitude terms (|d| - 1 ulp for d > 0, |d| + 1 ulp for d < 0). The reference's pins carry exactly that shift.
static float printed_sort_key_distance(float dist) { return -int64_t_to_float(-float_to_int64_t(dist)); }

static FILE* g_pin_dump = nullptr;       // --dump-vector-pins: write tests/golden/vector_pins.json for the GPU / emulator tiers
static void dump_matrix(const char* name, const std::vector<std::vector<float>>& rows, bool last = false) {
    if (!g_pin_dump) return;
    fprintf(g_pin_dump, "  \"%s\": [", name);
    for (size_t i = 0; i < rows.size(); i++) {
        fprintf(g_pin_dump, "%s[", i ? ", " : "");
        for (size_t j = 0; j < rows[i].size(); j++) fprintf(g_pin_dump, "%s%.9g", j ? ", " : "", rows[i][j]);
        fprintf(g_pin_dump, "]");
    }
    fprintf(g_pin_dump, "]%s\n", last ? "" : ",");
}

static void test_vector_reference_pins() {
    const std::vector<float> q1 = {std::stof("0.96826"), std::stof("0.94"), std::stof("0.39557"), std::stof("0.306488")};
    {   // CollectionVectorTest.BasicVectorQuerying, test/collection_vector_search_test.cpp:75-137 (cosine, 4-d)
        Index idx(1, 1);
        idx.vec_init(4, cosine);
        std::vector<std::vector<float>> values = {{0.851758f, 0.909671f, 0.823431f, 0.372063f}, {0.97826f, 0.933157f, 0.39557f, 0.306488f},
                                                  {0.230606f, 0.634397f, 0.514009f, 0.399594f}};
        for (uint32_t i = 0; i < 3; i++) idx.vec_add(i, values[i].data());
        vector_query_t vq; vq.values = q1;
        auto r = idx.search_vector(vq, {{SORT_VECTOR_DISTANCE, 0, -1}, {SORT_SEQ_ID, 0, 1}}, 10);
        CHECK_EQ(r.kvs.size(), 3u);                                                       // :113-114 found 3
        CHECK_EQ(r.kvs[0].key, 1u); CHECK_EQ(r.kvs[1].key, 0u); CHECK_EQ(r.kvs[2].key, 2u);   // :116-118
        CHECK_PIN_BITS(r.kvs[0].vector_distance, 3.409385681152344e-05);                  // :120-122 (bit-exact here)
        CHECK_PIN_BITS(r.kvs[1].vector_distance, 0.04329806566238403);
        CHECK_PIN_BITS(r.kvs[2].vector_distance, 0.15141665935516357);
        std::vector<uint32_t> filt = {0, 1};                                              // :124-137 points:[0,1]
        r = idx.search_vector(vq, {{SORT_VECTOR_DISTANCE, 0, -1}, {SORT_SEQ_ID, 0, 1}}, 10, &filt);
        CHECK_EQ(r.kvs.size(), 2u); CHECK_EQ(r.kvs[0].key, 1u); CHECK_EQ(r.kvs[1].key, 0u);
        dump_matrix("basic_docs", values);
    }
    {   // CollectionVectorTest.VecSearchWithFiltering, :806-901: 20 docs x 4, seed 47, uniform [0,1), cosine, flat path
        Index idx(1, 1);
        idx.vec_init(4, cosine);
        std::mt19937 rng; rng.seed(47);
        std::uniform_real_distribution<> distrib;
        std::vector<std::vector<float>> docs;
        for (size_t i = 0; i < 20; i++) {
            std::vector<float> values;
            for (size_t j = 0; j < 4; j++) values.push_back(distrib(rng));
            idx.vec_add((uint32_t)i, values.data());
            docs.push_back(values);
        }
        // the generator itself is pinned: BasicVectorQuerying's literals are documents 0..2 of this stream printed with 6 digits
        CHECK(std::fabs(docs[1][0] - 0.97826f) < 1e-6f && std::fabs(docs[1][3] - 0.306488f) < 1e-6f && std::fabs(docs[0][0] - 0.851758f) < 1e-6f);
        vector_query_t vq; vq.values = q1;
        std::vector<sort_by_t> sort = {{SORT_VECTOR_DISTANCE, 0, -1}, {SORT_SEQ_ID, 0, 1}};
        auto r = idx.search_vector(vq, sort, 20);
        CHECK_EQ(r.kvs.size(), 20u);                                                      // :848-849
        std::vector<uint32_t> filt; for (uint32_t i = 0; i < 10; i++) filt.push_back(i);  // points:<10
        // :851-863 `flat_search_cutoff: 0` with points:<10: the k-cut (HNSW-shaped) branch, fetch_size 20 -> found 10, 10 hits
        vq.flat_search_cutoff = 0;
        r = idx.search_vector(vq, sort, 20, &filt);
        CHECK_EQ(r.result_ids.size(), 10u); CHECK_EQ(r.kvs.size(), 10u);                  // :862-863
        // :865-881 `flat_search_cutoff: 1000`, per_page 3: the FLAT branch (10 filter ids < 1000): every filter id goes to the Topster
        // (capacity min(max(3, 250), 10) = 10, src/index.cpp:3506-3512), `found` = all 10 of them although only 3 hits are fetched
        vq.flat_search_cutoff = 1000;
        r = idx.search_vector(vq, sort, 3, &filt);
        CHECK_EQ(r.result_ids.size(), 10u);                                               // :874 ASSERT_EQ(10, results["found"])
        CHECK_EQ(r.num_keyword_matches, 10u);
        CHECK_EQ(r.kvs.size(), 10u);                                                      // (the response pages 3 of them, :875)
        CHECK_EQ(r.kvs[0].key, 1u); CHECK_PIN(r.kvs[0].vector_distance, 3.409385e-05);    // :877-878 ASSERT_FLOAT_EQ
        CHECK_EQ(r.kvs[1].key, 5u); CHECK_PIN(r.kvs[1].vector_distance, 0.016780376);     // :880-881
        CHECK(ulp_gap(r.kvs[0].vector_distance, 3.409385e-05f) <= 4 && ulp_gap(r.kvs[1].vector_distance, 0.016780376f) <= 4);
        for (size_t i = 1; i < r.kvs.size(); i++) CHECK(r.kvs[i - 1].vector_distance <= r.kvs[i].vector_distance);
        // the same request WITHOUT the flat branch cuts at k = fetch_size = 3: found would be 3 — the pin above tells the two apart
        vq.flat_search_cutoff = 0;
        CHECK_EQ(idx.search_vector(vq, sort, 3, &filt).result_ids.size(), 3u);
        // :883-901 `vec:([], id: 3, flat_search_cutoff: 1000)`: the query is document 3's STORED JSON values (src/vector_query_ops.cpp:137-146),
        // the document itself is dropped from the results (src/index.cpp:3651-3654, :3686); flat branch again
        vector_query_t vid; vid.values = docs[3]; vid.flat_search_cutoff = 1000; vid.query_doc_given = true; vid.seq_id = 3;
        r = idx.search_vector(vid, sort, 3, &filt);
        CHECK_EQ(r.result_ids.size(), 9u);                                                // the 10 filter ids minus document 3 itself
        CHECK(r.kvs.size() >= 3u);                                                        // :890 3 hits
        CHECK_EQ(r.kvs[0].key, 9u); CHECK_PIN(r.kvs[0].vector_distance, 0.050603985);     // :895-896
        CHECK_EQ(r.kvs[1].key, 5u); CHECK_PIN(r.kvs[1].vector_distance, 0.100155532);     // :898-899
        CHECK(ulp_gap(r.kvs[0].vector_distance, 0.050603985f) <= 4 && ulp_gap(r.kvs[1].vector_distance, 0.100155532f) <= 4);
        // ... and through the k-cut branch (k = 3 + 1 because document 3 passes the functor, :3651-3654): the same three hits
        vid.flat_search_cutoff = 0;
        r = idx.search_vector(vid, sort, 3, &filt);
        CHECK_EQ(r.kvs.size(), 3u); CHECK_EQ(r.kvs[0].key, 9u); CHECK_EQ(r.kvs[1].key, 5u);
        {   // duplicate embeddings: the flat branch's ties are the TOPSTER's (default sort [vector_distance asc, seq_id desc]: the larger
            // seq_id first, include/topster.h:146-154), the k-cut branch keeps hnswlib's smaller (distance, id) pairs at the cut
            Index dup(1, 1);
            dup.vec_init(4, cosine);
            for (uint32_t i = 0; i < 8; i++) dup.vec_add(i, docs[i % 2].data());            // ids 0,2,4,6 = docs[0]; 1,3,5,7 = docs[1]
            std::vector<uint32_t> f8 = {0, 1, 2, 3, 4, 5, 6, 7};
            vector_query_t vd; vd.values = q1; vd.flat_search_cutoff = 1000;
            dup.num_docs = 8;
            auto rf = dup.search_vector(vd, sort, 3, &f8);                                  // Topster capacity min(250, 8) = 8
            CHECK_EQ(rf.result_ids.size(), 8u); CHECK_EQ(rf.kvs.size(), 8u);
            CHECK_EQ(rf.kvs[0].key, 7u); CHECK_EQ(rf.kvs[1].key, 5u); CHECK_EQ(rf.kvs[2].key, 3u); CHECK_EQ(rf.kvs[3].key, 1u);   // docs[1] is nearer (pin above)
            CHECK_EQ(rf.kvs[4].key, 6u); CHECK_EQ(rf.kvs[7].key, 0u);
            vd.flat_search_cutoff = 0;
            auto rk = dup.search_vector(vd, sort, 3, &f8);                                  // k = 3: the cut keeps ids 1, 3, 5; the Topster orders them
            CHECK_EQ(rk.result_ids.size(), 3u);
            CHECK_EQ(rk.kvs[0].key, 5u); CHECK_EQ(rk.kvs[1].key, 3u); CHECK_EQ(rk.kvs[2].key, 1u);
        }
        dump_matrix("seed47_unit_docs", docs);
    }
    {   // CollectionVectorTest.TestDistanceThresholdWithIP, :5093-5196: 5 docs x 5, seed 47, uniform(-1,1) interleaved with
        // uniform_int(0,100) rank scores; IP metric; distance as a SORT KEY (`_vector_query(...)`): compute_sort_scores :5837-5851
        Index idx(1, 1);
        idx.vec_init(5, ip);
        std::mt19937 rng; rng.seed(47);
        std::uniform_real_distribution<> distrib(-1, 1);
        std::uniform_int_distribution<> distrib2(0, 100);
        std::vector<std::vector<float>> docs;
        std::vector<int> rank_score;
        for (int i = 0; i < 5; ++i) {
            std::vector<float> vector(5);
            std::generate(vector.begin(), vector.end(), [&]() { return distrib(rng); });
            rank_score.push_back(distrib2(rng));
            idx.vec_add((uint32_t)i, vector.data());
            docs.push_back(vector);
        }
        const std::vector<float> q = {std::stof("0.11731103425347378"), std::stof("-0.6694758317235057"), std::stof("-0.6211945774857595"),
                                      std::stof("-0.27966758971688255"), std::stof("-0.4683744007950299")};
        struct row { float printed; int rank; uint32_t id; };
        auto run = [&](const std::vector<float>& query, float threshold) {
            std::vector<row> rows;
            for (uint32_t i = 0; i < 5; i++) {
                float dist = Index::ip_distance(query.data(), idx.vec_get(i), 5);                     // :5842
                if (dist > threshold) dist = std::numeric_limits<float>::max();                         // :5844-5848
                rows.push_back({printed_sort_key_distance(dist), rank_score[i], i});
            }
            // sort_by _text_match:desc (all equal: one token, same field length... every doc matches "document"), distance asc, rank_score desc
            std::stable_sort(rows.begin(), rows.end(), [](const row& a, const row& b) { return a.printed < b.printed || (a.printed == b.printed && a.rank > b.rank); });
            return rows;
        };
        auto rows = run(q, 1.0f);
        // libstdc++'s uniform_int_distribution algorithm changed across releases; the rank scores only label the hits
        bool ranks_ok = rows[0].rank == 93 && rows[1].rank == 51 && rows[2].rank == 94 && rows[3].rank == 80 && rows[4].rank == 18;   // :5145-5154
        printf("vector pins: rank_score labels %s the reference's (this libstdc++'s uniform_int_distribution)\n", ranks_ok ? "reproduce" : "do NOT reproduce");
        CHECK_PIN_BITS(rows[0].printed, 0.2189185470342636);                              // :5146 ASSERT_EQ -> bit-exact
        CHECK_PIN_BITS(rows[1].printed, 0.7371898889541626);                              // :5148
        for (int i = 2; i < 5; i++) CHECK_PIN_BITS(rows[i].printed, 3.4028232635611926e+38);   // :5150-5154 FLT_MAX after the same shift
        // the RAW distances (what tsgpu returns) sit exactly one ulp from those printed values
        {
            float raw0 = Index::ip_distance(q.data(), idx.vec_get(rows[0].id), 5);
            CHECK_PIN(raw0, 0.2189185470342636); CHECK_EQ(ulp_gap(raw0, 0.2189185470342636f), 1);
        }
        const std::vector<float> qneg(5, -100.0f);                                        // :5177-5196, no threshold
        rows = run(qneg, std::numeric_limits<float>::max());
        const uint32_t want_id[5] = {1, 2, 4, 3, 0};
        const double want_d[5] = {-45.23314666748047, -38.66290283203125, -36.0988655090332, 9.637892723083496, 288.0364685058594};
        for (int i = 0; i < 5; i++) { CHECK_EQ(rows[i].id, want_id[i]); CHECK_PIN_BITS(rows[i].printed, want_d[i]); }
        for (int i = 0; i < 5; i++) { float raw = Index::ip_distance(qneg.data(), idx.vec_get(want_id[i]), 5); CHECK_PIN(raw, want_d[i]); CHECK_EQ(ulp_gap(raw, (float)want_d[i]), 1); }
        dump_matrix("seed47_ip_docs", docs);
        if (g_pin_dump) {
            fprintf(g_pin_dump, "  \"seed47_ip_rank_scores\": [%d, %d, %d, %d, %d],\n", rank_score[0], rank_score[1], rank_score[2], rank_score[3], rank_score[4]);
        }
    }
    printf("vector pins: max gap to the reference's pinned distances = %d ulp (raw distance; the printed sort-key form is bit-exact)\n", g_max_pin_ulp);
}

// ---------- hybrid rank fusion formula: test/collection_vector_search_test.cpp:1429-1431 ----------
static void test_hybrid() {
    Index idx(1, 1);
    idx.vec_init(4, ip);
    // 3 docs; text ranks via token adjacency, vector ranks via explicit vectors
    idx.index_plain_field(0, 0, {1, 2, 4});
    idx.index_plain_field(1, 0, {1, 4, 2});
    idx.index_plain_field(2, 0, {4, 4, 4});
    float v0[4] = {0, 0, 0, 1}, v1[4] = {1, 0, 0, 0}, v2[4] = {0.9f, 0, 0, 0};
    idx.vec_add(0, v0); idx.vec_add(1, v1); idx.vec_add(2, v2);
    keyword_query_t q;
    q.fields = {{0, 15}};
    q.sort = {{SORT_TEXT_MATCH, 0, 1}, {SORT_SEQ_ID, 0, 1}};
    q.tokens = {1, 2};
    vector_query_t vq;
    vq.values = {1, 0, 0, 0};
    auto r = idx.search_hybrid(q, vq);
    CHECK_EQ(r.kvs.size(), 3u);
    // text: doc0 rank1, doc1 rank2; vector: doc1 rank1 (dist 0), doc2 rank2 (0.1), doc0 rank3 (1.0)
    auto fused = [&](int tr, int vr) -> int64_t {   // index.cpp:4111 then :4156-4158 / :4179
        const float VW = 0.3f, TW = 1.0 - VW;
        if (!tr) return float_to_int64_t((1.0 / vr) * VW);
        int64_t text_part = float_to_int64_t((1.0 / tr) * TW);
        return float_to_int64_t(int64_t_to_float(text_part) + ((1.0 / vr) * VW));
    };
    for (auto& kv : r.kvs) {
        if (kv.key == 0) CHECK_EQ(kv.scores[0], fused(1, 3));
        if (kv.key == 1) CHECK_EQ(kv.scores[0], fused(2, 1));
        if (kv.key == 2) CHECK_EQ(kv.scores[0], fused(0, 2));
    }
    // 1/1*0.7 + 1/3*0.3 = 0.8 ; 1/2*0.7 + 1/1*0.3 = 0.65 ; 1/2*0.3 = 0.15
    CHECK_EQ(r.kvs[0].key, 0u); CHECK_EQ(r.kvs[1].key, 1u); CHECK_EQ(r.kvs[2].key, 2u);
    CHECK(std::fabs(int64_t_to_float(r.kvs[0].scores[0]) - 0.8f) < 1e-6f);
    CHECK(std::fabs(int64_t_to_float(r.kvs[1].scores[0]) - 0.65f) < 1e-6f);
    CHECK(std::fabs(int64_t_to_float(r.kvs[2].scores[0]) - 0.15f) < 1e-6f);
    CHECK_EQ(r.kvs[2].text_match_score, 0);
    CHECK(r.kvs[2].vector_distance > 0.09f && r.kvs[2].vector_distance < 0.11f);
}

int main(int argc, char** argv) {
    if (argc > 1) g_golden_dir = argv[1];
    if (argc > 3 && std::string(argv[2]) == "--dump-vector-pins") { g_pin_dump = fopen(argv[3], "w"); if (g_pin_dump) fprintf(g_pin_dump, "{\n"); }
    test_arrays();
    test_posting_lists();
    test_or_iterator();
    test_match();
    test_topster();
    test_distinct_topster();
    test_text_scores();
    test_vectors();
    test_vector_reference_pins();
    test_hybrid();
    if (g_pin_dump) { fprintf(g_pin_dump, "  \"generator\": \"oracle/golden_tests.cpp --dump-vector-pins (std::mt19937 seed 47, the reference's test generators)\"\n}\n"); fclose(g_pin_dump); }
    printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail == 0 ? 0 : 1;
}
