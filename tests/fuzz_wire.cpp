// CPU-only sanitizer fuzz of the wire / bag parsers (tests/test_wire.py builds it with g++ -fsanitize=address,undefined together
// with csrc/uzl_wire.hip, which is plain host C++): mutated and truncated messages must never read out of bounds
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <fstream>
#include <iterator>
#include "../include/uzl_mi355x.h"
static std::vector<uint8_t> rd(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {}); }
static uint32_t s = 1; static uint32_t rnd() { s = s * 1664525u + 1013904223u; return s >> 8; }
int main(int argc, char** argv)
{
    std::vector<std::vector<uint8_t>> seeds; for (int i = 1; i < argc; i++) seeds.push_back(rd(argv[i]));
    long ok[4] = {0, 0, 0, 0}, n = 0;
    for (int it = 0; it < 100000; it++) {
        std::vector<uint8_t> m = seeds[rnd() % seeds.size()];
        const int muts = rnd() % 4;
        for (int k = 0; k < muts && !m.empty(); k++) {
            const size_t pos = rnd() % m.size();
            switch (rnd() % 4) { case 0: m[pos] = (uint8_t)rnd(); break; case 1: m[pos] ^= 1u << (rnd() % 8); break;
                case 2: { uint32_t v = (rnd() % 3 == 0) ? 0xffffffffu : rnd() % 100000; if (pos + 4 <= m.size()) memcpy(&m[pos], &v, 4); break; }
                case 3: m.resize(pos); break; }
        }
        // exact-size heap copy so that any overread trips the sanitizer
        uint8_t* b = (uint8_t*)malloc(m.size() ? m.size() : 1); memcpy(b, m.data(), m.size());
        uzl_wire_edge e; uint64_t used = 0;
        if (uzl_wire_edge_decode(b, m.size(), &e, &used) == UZL_OK) { ok[0]++; std::vector<uint8_t> o(uzl_wire_edge_size(&e)); uint64_t w; uzl_wire_edge_encode(&e, o.data(), o.size(), &w); }
        uzl_wire_node nd; uzl_wire_sensor sens[4]; uzl_span eids[4]; int64_t st[4];
        if (uzl_wire_node_decode(b, m.size(), &nd, 4, st, 4, eids, 4, sens, &used) == UZL_OK) {
            ok[1]++;
            volatile char acc = 0;
            for (int i = 0; i < nd.n_sensors && i < 4; i++) { if (sens[i].records.n) acc += sens[i].records.p[sens[i].records.n - 1]; if (sens[i].raw.n) acc += sens[i].raw.p[sens[i].raw.n - 1]; }
            for (int i = 0; i < nd.n_edge_ids && i < 4; i++) if (eids[i].n) acc += eids[i].p[eids[i].n - 1];
        }
        uzl_wire_meta mt; uzl_wire_sensor_transform tr[3], tri[3];          // (capacities below the counts a mutated message may claim)
        if (uzl_wire_meta_decode(b, m.size(), &mt, 3, tr, 3, tri, &used) == UZL_OK) {
            ok[3]++;
            volatile char acc = 0;
            if (mt.name.n) acc += mt.name.p[mt.name.n - 1];
            if (mt.frame_id.n) acc += mt.frame_id.p[mt.frame_id.n - 1];
            for (int i = 0; i < mt.n_sensor_transforms && i < 3; i++) if (tr[i].sensor_name.n) acc += tr[i].sensor_name.p[tr[i].sensor_name.n - 1];
            for (int i = 0; i < mt.n_sensor_transforms_initial && i < 3; i++) if (tri[i].sensor_name.n) acc += tri[i].sensor_name.p[tri[i].sensor_name.n - 1];
            if (mt.n_sensor_transforms <= 3 && mt.n_sensor_transforms_initial <= 3) {
                std::vector<uint8_t> o(uzl_wire_meta_size(&mt, tr, tri)); uint64_t w; uzl_wire_meta_encode(&mt, tr, tri, o.data(), o.size(), &w);
            }
        }
        uzl_bag_msg msgs[4]; int32_t nm = 0;
        if (uzl_bag_read(b, m.size(), 4, msgs, &nm) == UZL_OK) { ok[2]++; volatile char acc = 0; for (int i = 0; i < nm && i < 4; i++) { if (msgs[i].data.n) acc += msgs[i].data.p[msgs[i].data.n - 1]; if (msgs[i].topic.n) acc += msgs[i].topic.p[0]; } }
        free(b); n++;
    }
    printf("fuzz: %ld inputs, decoded ok: edge %ld node %ld bag %ld meta %ld\n", n, ok[0], ok[1], ok[2], ok[3]);
    return 0;
}
