// entropy.h -- the byte-level coding tools of the ETC1S backend: length-limited canonical Huffman tables, the LSB-first bit writer
// and the serialised form of a Huffman table. Host code; the output has to be byte-identical to the reference's because the
// transcoder on the other side parses it (transcoder/basisu_transcoder_internal.h:636-760).
//
//   huffman_table::init   = huffman_encoding_table::init            encoder/basisu_enc.cpp:1426-1529
//   bit_writer            = bitwise_coder                           encoder/basisu_enc.h:2466-2670
//   bit_writer::put_table = bitwise_coder::emit_huffman_table       encoder/basisu_enc.cpp:1531-1661
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace bu {

enum : uint32_t {
    kHuffMaxSyms = 1u << 14, kHuffMaxSymsLog2 = 14, kHuffMaxCodeSize = 16, kHuffMaxInternalCodeSize = 31,  // transcoder/basisu.h:489-491
    kHuffCodelengthCodes = 21, kHuffSmallZeroRun = 17, kHuffBigZeroRun = 18, kHuffSmallRepeat = 19, kHuffBigRepeat = 20  // basisu.h:494-505
};

class huffman_table {
public:
    // freq[n]: occurrence counts. Lengths are those of the Huffman tree that merges, on equal weights, a leaf before an internal node
    // (the tie rule of the in-place Moffat-Katajainen construction the reference uses), handed out along the (count, symbol)-sorted
    // order, then squeezed to max_code_size by the reference's Kraft-sum repair.
    bool init(const uint32_t* freq, uint32_t n, uint32_t max_code_size = kHuffMaxCodeSize) {
        m_sizes.assign(n, 0);
        m_codes.assign(n, 0);
        if (!n || n > kHuffMaxSyms || max_code_size > kHuffMaxCodeSize) return false;
        // counts are scaled into 16 bits when the largest does not fit (enc.cpp:1507-1526)
        uint32_t max_freq = 0;
        for (uint32_t i = 0; i < n; i++) max_freq = std::max(max_freq, freq[i]);
        struct leaf { uint32_t weight, sym; };
        std::vector<leaf> leaves;
        for (uint32_t i = 0; i < n; i++) {
            if (!freq[i]) continue;
            uint32_t w = freq[i];
            if (max_freq >= 0xFFFFu) {
                w = (uint32_t)(((uint64_t)freq[i] * 65534u + (max_freq >> 1)) / max_freq);
                w = std::min(std::max(w, 1u), 65534u);
            }
            leaves.push_back(leaf{w, i});
        }
        const uint32_t used = (uint32_t)leaves.size();
        if (!used) return false;
        std::stable_sort(leaves.begin(), leaves.end(), [](const leaf& a, const leaf& b) { return a.weight < b.weight; });

        uint32_t per_len[kHuffMaxInternalCodeSize + 2] = {0};
        if (used == 1) {
            per_len[1] = 1;
        } else {
            // two-queue construction: leaves ascending, internal nodes in creation order (their weights are non-decreasing)
            const uint32_t internal = used - 1;
            std::vector<uint64_t> iw(internal);
            std::vector<uint32_t> leaf_parent(used), node_parent(internal, 0);
            uint32_t lq = 0, iq = 0;
            for (uint32_t made = 0; made < internal; made++) {
                uint64_t w = 0;
                for (int pick = 0; pick < 2; pick++) {
                    const bool take_internal = (lq >= used) || (iq < made && iw[iq] < leaves[lq].weight);
                    if (take_internal) { w += iw[iq]; node_parent[iq++] = made; }
                    else { w += leaves[lq].weight; leaf_parent[lq++] = made; }
                }
                iw[made] = w;
            }
            std::vector<uint32_t> depth(internal, 0);  // the last node made is the root
            for (uint32_t k = internal - 1; k-- > 0;) depth[k] = depth[node_parent[k]] + 1;
            for (uint32_t l = 0; l < used; l++) {
                const uint32_t d = depth[leaf_parent[l]] + 1;
                if (d > kHuffMaxInternalCodeSize) return false;
                per_len[d]++;
            }
            // length limit (enc.cpp:1354-1382): fold the overlong codes into the limit, then repair the Kraft sum one unit at a time
            for (uint32_t i = max_code_size + 1; i <= kHuffMaxInternalCodeSize; i++) { per_len[max_code_size] += per_len[i]; per_len[i] = 0; }
            uint64_t kraft = 0;
            for (uint32_t i = max_code_size; i >= 1; i--) kraft += (uint64_t)per_len[i] << (max_code_size - i);
            while (kraft != (1ull << max_code_size)) {
                per_len[max_code_size]--;
                for (uint32_t i = max_code_size - 1; i >= 1; i--)
                    if (per_len[i]) { per_len[i]--; per_len[i + 1] += 2; break; }
                kraft--;
            }
        }
        // shortest codes to the back of the sorted order
        uint32_t j = used;
        for (uint32_t len = 1; len <= max_code_size; len++)
            for (uint32_t c = per_len[len]; c > 0; c--) m_sizes[leaves[--j].sym] = (uint8_t)len;
        // canonical codes, stored bit-reversed because the writer is LSB first
        uint32_t next_code[kHuffMaxCodeSize + 2] = {0};
        for (uint32_t len = 2, c = 0; len <= max_code_size; len++) next_code[len] = c = (c + per_len[len - 1]) << 1;
        for (uint32_t s = 0; s < n; s++) {
            const uint32_t len = m_sizes[s];
            if (!len) continue;
            uint32_t code = next_code[len]++, rev = 0;
            for (uint32_t b = 0; b < len; b++, code >>= 1) rev = (rev << 1) | (code & 1u);
            m_codes[s] = (uint16_t)rev;
        }
        return true;
    }
    bool init(const std::vector<uint32_t>& hist, uint32_t max_code_size = kHuffMaxCodeSize) { return init(hist.data(), (uint32_t)hist.size(), max_code_size); }

    const std::vector<uint8_t>& sizes() const { return m_sizes; }
    const std::vector<uint16_t>& codes() const { return m_codes; }
    uint32_t total_used() const {  // index of the last coded symbol + 1
        for (size_t i = m_sizes.size(); i > 0; i--) if (m_sizes[i - 1]) return (uint32_t)i;
        return 0;
    }

private:
    std::vector<uint8_t> m_sizes;
    std::vector<uint16_t> m_codes;
};

class bit_writer {
public:
    void restart(size_t reserve_bytes = 1024) { m_bytes.clear(); m_bytes.reserve(reserve_bytes); m_acc = 0; m_fill = 0; }
    uint32_t put_bits(uint32_t bits, uint32_t n) {
        if (!n) return 0;
        m_acc |= (uint64_t)bits << m_fill;
        m_fill += n;
        while (m_fill >= 8) { m_bytes.push_back((uint8_t)m_acc); m_acc >>= 8; m_fill -= 8; }
        return n;
    }
    uint32_t put_code(uint32_t sym, const huffman_table& t) { return put_bits(t.codes()[sym], t.sizes()[sym]); }
    uint32_t put_vlc(uint32_t v, uint32_t chunk_bits) {  // enc.h:2640-2661: chunks, low first, each with a continuation flag on top
        uint32_t total = 0;
        for (;;) {
            const uint32_t rest = v >> chunk_bits;
            total += put_bits((v & ((1u << chunk_bits) - 1u)) | (rest ? (1u << chunk_bits) : 0u), chunk_bits + 1);
            if (!rest) return total;
            v = rest;
        }
    }
    void flush() { if (m_fill) { m_bytes.push_back((uint8_t)m_acc); m_acc = 0; m_fill = 0; } }
    const std::vector<uint8_t>& bytes() const { return m_bytes; }
    // Everything `other` holds (whole bytes and its unflushed tail) behind what this writer holds, as if it had been put here bit by bit: lets one long symbol
    // stream be coded in pieces on several threads and joined afterwards.
    void append(const bit_writer& other) {
        const std::vector<uint8_t>& ob = other.m_bytes;
        if (!m_fill) {
            m_bytes.insert(m_bytes.end(), ob.begin(), ob.end());
        } else {
            const size_t at = m_bytes.size(), n = ob.size();
            m_bytes.resize(at + n);
            uint8_t* dst = m_bytes.data() + at;
            uint32_t carry = (uint32_t)m_acc;   // m_fill bits
            const uint32_t sh = m_fill;
            for (size_t i = 0; i < n; i++) { const uint32_t v = carry | ((uint32_t)ob[i] << sh); dst[i] = (uint8_t)v; carry = v >> 8; }
            m_acc = carry;
        }
        if (other.m_fill) put_bits((uint32_t)other.m_acc, other.m_fill);
    }

    // A table travels as its code lengths, run-length tokenised (literal 0..16, short/long zero runs, short/long repeats of the
    // previous length) and coded with a 21-symbol table of at most 7-bit codes whose own lengths go first in a fixed order.
    bool put_table(const huffman_table& t) {
        const uint32_t used = t.total_used();
        put_bits(used, kHuffMaxSymsLog2);
        if (!used) return true;
        std::vector<uint16_t> tok;  // low 6 bits: code, rest: the extra-bits value
        auto zeros = [&](uint32_t run) {
            while (run) {
                const uint32_t r = std::min(run, 138u);
                if (r < 3) for (uint32_t k = 0; k < r; k++) tok.push_back(0);
                else if (r <= 10) tok.push_back((uint16_t)(kHuffSmallZeroRun | ((r - 3) << 6)));
                else tok.push_back((uint16_t)(kHuffBigZeroRun | ((r - 11) << 6)));
                run -= r;
            }
        };
        auto repeats = [&](uint32_t len, uint32_t run) {  // run = occurrences after the literal
            while (run) {
                const uint32_t r = std::min(run, 134u);
                if (r < 3) for (uint32_t k = 0; k < r; k++) tok.push_back((uint16_t)len);
                else if (r <= 6) tok.push_back((uint16_t)(kHuffSmallRepeat | ((r - 3) << 6)));
                else tok.push_back((uint16_t)(kHuffBigRepeat | ((r - 7) << 6)));
                run -= r;
            }
        };
        const std::vector<uint8_t>& sizes = t.sizes();
        for (uint32_t i = 0; i < used;) {
            uint32_t e = i + 1;
            while (e < used && sizes[e] == sizes[i]) e++;
            if (!sizes[i]) zeros(e - i);
            else { tok.push_back(sizes[i]); repeats(sizes[i], e - i - 1); }
            i = e;
        }
        std::vector<uint32_t> h(kHuffCodelengthCodes, 0);
        for (uint16_t v : tok) h[v & 63]++;
        huffman_table ct;
        if (!ct.init(h, 7)) return false;
        static const uint8_t order[kHuffCodelengthCodes] = {17, 18, 19, 20, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15, 16};  // basisu.h:508
        uint32_t sent = kHuffCodelengthCodes;
        while (sent && !ct.sizes()[order[sent - 1]]) sent--;
        put_bits(sent, 5);
        for (uint32_t i = 0; i < sent; i++) put_bits(ct.sizes()[order[i]], 3);
        for (uint16_t v : tok) {
            const uint32_t code = v & 63, extra = v >> 6;
            put_code(code, ct);
            if (code == kHuffSmallZeroRun) put_bits(extra, 3);
            else if (code == kHuffBigZeroRun) put_bits(extra, 7);
            else if (code == kHuffSmallRepeat) put_bits(extra, 2);
            else if (code == kHuffBigRepeat) put_bits(extra, 7);
        }
        return true;
    }

private:
    std::vector<uint8_t> m_bytes;
    uint64_t m_acc = 0;
    uint32_t m_fill = 0;
};

}  // namespace bu
