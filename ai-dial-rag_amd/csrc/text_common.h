// Host-side string helpers shared by stem_english.cpp and keywords_preprocess.cpp (no device code).
#pragma once
#include <cstddef>
#include <string>

namespace mir {
namespace text {

using U = std::u32string;

// Snowball-English stem of one LOWER-CASED token, in place (stem_english.cpp; r1 / r2 are scratch)
void snowball_english(U &word, U &r1, U &r2);
// UTF-8 <-> code points; malformed bytes pass through as single code points >= 0x110000 and come back as they were
void utf8_decode(const unsigned char *p, size_t n, U &out);
size_t utf8_encode(const char32_t *w, size_t n, unsigned char *dst);

}  // namespace text
}  // namespace mir
