// fmx_text.cc — restricted text front-end (see fmx_text.hh for scope and pinning).
#include "fmx_text.hh"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

namespace fmx {
namespace {

// ---------------------------------------------------------------------------------------------
// normalize() — src/fuzzy_match.cc:43-58: icu::Normalizer2::getNFCInstance()->normalize(UnicodeString::fromUTF8(text)),
// back to UTF-8; on an ICU error the text is returned unchanged.  ICU is a system library, not part of this build: its
// C API (unorm2_*, u_strFromUTF8WithSub, u_strToUTF8WithSub — what the C++ calls above are made of) is bound at run time
// from the libicuuc the box has.  Symbols carry the library's major version as a suffix ("unorm2_normalize_70").
// ---------------------------------------------------------------------------------------------
typedef uint16_t UCh;
struct Icu {
  const void* (*get_nfc)(int* err) = nullptr;
  int32_t (*normalize)(const void* n2, const UCh* src, int32_t len, UCh* dst, int32_t cap, int* err) = nullptr;
  UCh* (*from_utf8)(UCh* dst, int32_t cap, int32_t* out_len, const char* src, int32_t len, int32_t sub, int32_t* n_sub, int* err) = nullptr;
  char* (*to_utf8)(char* dst, int32_t cap, int32_t* out_len, const UCh* src, int32_t len, int32_t sub, int32_t* n_sub, int* err) = nullptr;
  const void* nfc = nullptr;
  bool ok = false;
};

const Icu& icu() {
  static Icu g;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    int ver = 0;
    if (const char* path = getenv("FMX_ICU_LIB")) {  // e.g. /usr/lib/x86_64-linux-gnu/libicuuc.so.70 ("none" disables)
      if (strcmp(path, "none") != 0) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      if (h) { const char* dot = strrchr(path, '.'); if (dot) ver = atoi(dot + 1); }
    } else {
      for (int v = 80; v >= 50 && !h; v--) {
        char name[64];
        snprintf(name, sizeof name, "libicuuc.so.%d", v);
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) ver = v;
      }
    }
    auto sym = [&](const char* base) -> void* {
      char name[96];
      snprintf(name, sizeof name, "%s_%d", base, ver);
      void* p = h ? dlsym(h, name) : nullptr;
      if (!p && h) p = dlsym(h, base);  // builds without symbol renaming
      return p;
    };
    if (h) {
      g.get_nfc = (decltype(g.get_nfc))sym("unorm2_getNFCInstance");
      g.normalize = (decltype(g.normalize))sym("unorm2_normalize");
      g.from_utf8 = (decltype(g.from_utf8))sym("u_strFromUTF8WithSub");
      g.to_utf8 = (decltype(g.to_utf8))sym("u_strToUTF8WithSub");
      if (g.get_nfc && g.normalize && g.from_utf8 && g.to_utf8) {
        int err = 0;
        g.nfc = g.get_nfc(&err);
        g.ok = err <= 0 && g.nfc != nullptr;  // U_FAILURE(code) = code > U_ZERO_ERROR
      }
    }
    if (!g.ok)
      fprintf(stderr, "fmx: WARNING: no usable ICU (libicuuc) found: sentences are NOT NFC-normalised, input is taken as NFC "
                      "(the reference normalises, src/fuzzy_match.cc:43-58)\n");
  });
  return g;
}

std::string nfc_normalize(const std::string& text) {
  bool ascii = true;
  for (unsigned char c : text) if (c >= 0x80) { ascii = false; break; }
  if (ascii) return text;  // ASCII is NFC
  const Icu& u = icu();
  if (!u.ok) return text;
  int err = 0;
  int32_t n16 = 0, n_sub = 0;
  std::basic_string<UCh> src(text.size() + 1, 0);  // UTF-16 is never longer (in units) than UTF-8 (in bytes)
  u.from_utf8(&src[0], (int32_t)src.size(), &n16, text.data(), (int32_t)text.size(), 0xFFFD, &n_sub, &err);  // fromUTF8: ill-formed -> U+FFFD
  if (err > 0) return text;
  std::basic_string<UCh> dst((size_t)n16 * 3 + 16, 0);  // NFC grows a string at most threefold
  err = 0;
  const int32_t m16 = u.normalize(u.nfc, src.data(), n16, &dst[0], (int32_t)dst.size(), &err);
  if (err > 0) return text;  // "if (U_FAILURE(error_code)) return text_utf8"
  std::string out((size_t)m16 * 3 + 4, '\0');
  int32_t n8 = 0;
  err = 0;
  u.to_utf8(&out[0], (int32_t)out.size(), &n8, dst.data(), m16, 0xFFFD, &n_sub, &err);
  if (err > 0) return text;
  out.resize((size_t)n8);
  return out;
}

typedef uint32_t cp_t;

const char* const kJoiner = "\xEF\xBF\xAD";     // ￭ U+FFED  onmt::Tokenizer::joiner_marker
const char* const kSpacer = "\xE2\x96\x81";     // ▁ U+2581  onmt::Tokenizer::spacer_marker
const char* const kPhOpen = "\xEF\xBD\x9F";     // ｟ U+FF5F  onmt::Tokenizer::ph_marker_open
const char* const kPhClose = "\xEF\xBD\xA0";    // ｠ U+FF60  onmt::Tokenizer::ph_marker_close
const cp_t kPhOpenCp = 0xFF5F, kPhCloseCp = 0xFF60;

inline bool in(cp_t v, cp_t a, cp_t b) { return v >= a && v <= b; }

// one code point; malformed bytes decode as themselves (one byte)
cp_t decode(const unsigned char* s, size_t n, size_t& len) {
  const unsigned char c = s[0];
  len = 1;
  if (c < 0x80) return c;
  int need = (c >= 0xF0 && c < 0xF8) ? 3 : (c >= 0xE0) ? 2 : (c >= 0xC2 && c < 0xE0) ? 1 : 0;
  if (!need || c >= 0xF8 || (size_t)need >= n) return c;
  cp_t v = need == 1 ? (c & 0x1Fu) : need == 2 ? (c & 0x0Fu) : (c & 0x07u);
  for (int k = 1; k <= need; k++) {
    if ((s[k] & 0xC0) != 0x80) return c;
    v = (v << 6) | (s[k] & 0x3Fu);
  }
  len = (size_t)need + 1;
  return v;
}

void append_utf8(std::string& out, cp_t c) {
  if (c < 0x80) out += (char)c;
  else if (c < 0x800) { out += (char)(0xC0 | (c >> 6)); out += (char)(0x80 | (c & 0x3F)); }
  else if (c < 0x10000) { out += (char)(0xE0 | (c >> 12)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
  else { out += (char)(0xF0 | (c >> 18)); out += (char)(0x80 | ((c >> 12) & 0x3F)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
}

// onmt::unicode::is_separator: white space
bool is_separator(cp_t v) {
  return in(v, 9, 13) || v == 0x20 || v == 0x85 || v == 0xA0 || v == 0x1680 || in(v, 0x2000, 0x200A) || v == 0x2028 ||
         v == 0x2029 || v == 0x202F || v == 0x205F || v == 0x3000;
}

// combining marks (they stay attached to the character in front of them)
bool is_mark(cp_t v) {
  return in(v, 0x0300, 0x036F) || in(v, 0x0483, 0x0489) || in(v, 0x0591, 0x05BD) || v == 0x05BF || in(v, 0x05C1, 0x05C2) ||
         in(v, 0x05C4, 0x05C5) || v == 0x05C7 || in(v, 0x0610, 0x061A) || in(v, 0x064B, 0x065F) || v == 0x0670 ||
         in(v, 0x06D6, 0x06DC) || in(v, 0x06DF, 0x06E4) || in(v, 0x06E7, 0x06E8) || in(v, 0x06EA, 0x06ED) ||
         in(v, 0x0900, 0x0903) || in(v, 0x093A, 0x093C) || in(v, 0x093E, 0x094F) || in(v, 0x0951, 0x0957) || in(v, 0x0962, 0x0963) ||
         v == 0x0E31 || in(v, 0x0E34, 0x0E3A) || in(v, 0x0E47, 0x0E4E) || v == 0x0EB1 || in(v, 0x0EB4, 0x0EBC) || in(v, 0x0EC8, 0x0ECD) ||
         in(v, 0x102B, 0x103E) || in(v, 0x1AB0, 0x1AFF) || in(v, 0x1DC0, 0x1DFF) || in(v, 0x20D0, 0x20FF) || in(v, 0x3099, 0x309A) ||
         in(v, 0xFE00, 0xFE0F) || in(v, 0xFE20, 0xFE2F);
}

// onmt::unicode::is_number: decimal digits, letter numbers and other numbers of the covered blocks
bool is_number(cp_t v) {
  if (v < 0x80) return in(v, '0', '9');
  return v == 0xB2 || v == 0xB3 || v == 0xB9 || in(v, 0xBC, 0xBE) || in(v, 0x0660, 0x0669) || in(v, 0x06F0, 0x06F9) ||
         in(v, 0x07C0, 0x07C9) || in(v, 0x0966, 0x096F) || in(v, 0x09E6, 0x09EF) || in(v, 0x09F4, 0x09F9) || in(v, 0x0A66, 0x0A6F) ||
         in(v, 0x0AE6, 0x0AEF) || in(v, 0x0B66, 0x0B6F) || in(v, 0x0B72, 0x0B77) || in(v, 0x0BE6, 0x0BF2) || in(v, 0x0C66, 0x0C6F) ||
         in(v, 0x0C78, 0x0C7E) || in(v, 0x0CE6, 0x0CEF) || in(v, 0x0D66, 0x0D78) || in(v, 0x0DE6, 0x0DEF) || in(v, 0x0E50, 0x0E59) ||
         in(v, 0x0ED0, 0x0ED9) || in(v, 0x0F20, 0x0F33) || in(v, 0x1040, 0x1049) || in(v, 0x1090, 0x1099) || in(v, 0x1369, 0x137C) ||
         in(v, 0x17E0, 0x17E9) || in(v, 0x17F0, 0x17F9) || in(v, 0x1810, 0x1819) || v == 0x2070 || in(v, 0x2074, 0x2079) ||
         in(v, 0x2080, 0x2089) || in(v, 0x2150, 0x2182) || in(v, 0x2185, 0x2189) || in(v, 0x2460, 0x249B) || in(v, 0x24EA, 0x24FF) ||
         in(v, 0x2776, 0x2793) || v == 0x3007 || in(v, 0x3021, 0x3029) || in(v, 0x3038, 0x303A) || in(v, 0x3192, 0x3195) ||
         in(v, 0x3220, 0x3229) || in(v, 0x3248, 0x324F) || in(v, 0x3251, 0x325F) || in(v, 0x3280, 0x3289) || in(v, 0x32B1, 0x32BF) ||
         in(v, 0xFF10, 0xFF19);
}

// scripts the tokenizer tells apart (SegmentAlphabetChange); kNoScript for letters outside the covered blocks
enum Script : int {
  kNoScript = -1, kLatin, kGreek, kCyrillic, kArmenian, kHebrew, kArabic, kDevanagari, kBengali, kGurmukhi, kGujarati, kOriya,
  kTamil, kTelugu, kKannada, kMalayalam, kSinhala, kThai, kLao, kTibetan, kMyanmar, kGeorgian, kHangul, kEthiopic, kKhmer,
  kMongolian, kHiragana, kKatakana, kBopomofo, kKanbun, kHan,
};
const int kPlaceholderAlphabet = -2;

// letter (Unicode L*, with the marks of its block) of a covered block -> its script; otherwise "not a letter"
bool letter_script(cp_t v, int& script) {
  script = kNoScript;
  if (v < 0x80) { if (in(v, 'a', 'z') || in(v, 'A', 'Z')) { script = kLatin; return true; } return false; }
  if (is_number(v)) return false;
  if (v == 0xAA || v == 0xBA || (in(v, 0xC0, 0x24F) && v != 0xD7 && v != 0xF7) || in(v, 0x250, 0x2AF) || in(v, 0x1D00, 0x1DBF) ||
      in(v, 0x1E00, 0x1EFF) || in(v, 0x2C60, 0x2C7F) || in(v, 0xA720, 0xA7FF) || in(v, 0xAB30, 0xAB6F) || in(v, 0xFB00, 0xFB06) ||
      in(v, 0xFF21, 0xFF3A) || in(v, 0xFF41, 0xFF5A)) { script = kLatin; return true; }
  if (v == 0xB5) { script = kGreek; return true; }  // µ MICRO SIGN: a (common-script) lower-case letter
  if (in(v, 0x2B0, 0x2C1) || in(v, 0x2C6, 0x2D1) || in(v, 0x2E0, 0x2E4) || v == 0x2EC || v == 0x2EE) return true;  // modifier letters
  if ((in(v, 0x370, 0x3FF) && v != 0x375 && v != 0x37E && v != 0x384 && v != 0x385 && v != 0x387 && v != 0x3F6 && !in(v, 0x378, 0x379) &&
       !in(v, 0x380, 0x383)) || in(v, 0x1F00, 0x1FBC) || v == 0x1FBE || in(v, 0x1FC2, 0x1FCC) || in(v, 0x1FD0, 0x1FDB) ||
      in(v, 0x1FE0, 0x1FEC) || in(v, 0x1FF2, 0x1FFC)) { script = kGreek; return true; }
  if ((in(v, 0x400, 0x52F) && v != 0x482) || in(v, 0x1C80, 0x1C88) || in(v, 0x2DE0, 0x2DFF) || in(v, 0xA640, 0xA69F)) { script = kCyrillic; return true; }
  if (in(v, 0x531, 0x556) || v == 0x559 || in(v, 0x561, 0x587)) { script = kArmenian; return true; }
  if (in(v, 0x591, 0x5C7) && is_mark(v)) { script = kHebrew; return true; }
  if (in(v, 0x5D0, 0x5EA) || in(v, 0x5F0, 0x5F2) || in(v, 0xFB1D, 0xFB4F)) { script = kHebrew; return true; }
  if (in(v, 0x610, 0x61A) || in(v, 0x620, 0x65F) || in(v, 0x66E, 0x6D3) || in(v, 0x6D5, 0x6DC) || in(v, 0x6DF, 0x6E8) ||
      in(v, 0x6EA, 0x6EF) || in(v, 0x6FA, 0x6FC) || v == 0x6FF || in(v, 0x750, 0x77F) || in(v, 0x8A0, 0x8FF) || in(v, 0xFB50, 0xFBB1) ||
      in(v, 0xFBD3, 0xFD3D) || in(v, 0xFD50, 0xFDFB) || in(v, 0xFE70, 0xFEFC)) { script = kArabic; return true; }
  struct Blk { cp_t a, b; int s; };
  static const Blk indic[] = {{0x900, 0x97F, kDevanagari}, {0x980, 0x9FF, kBengali}, {0xA00, 0xA7F, kGurmukhi}, {0xA80, 0xAFF, kGujarati},
                              {0xB00, 0xB7F, kOriya}, {0xB80, 0xBFF, kTamil}, {0xC00, 0xC7F, kTelugu}, {0xC80, 0xCFF, kKannada},
                              {0xD00, 0xD7F, kMalayalam}, {0xD80, 0xDFF, kSinhala}};
  for (const Blk& k : indic)
    if (in(v, k.a, k.b)) {
      if (v == 0x964 || v == 0x965 || v == 0x970 || v == 0x9FD || v == 0xA76 || v == 0xAF0 || v == 0xDF4 || in(v, 0x9F2, 0x9FB) ||
          in(v, 0xBF3, 0xBFA) || v == 0xC7F || v == 0xD4F || v == 0xD79) return false;  // danda, abbreviation / currency signs
      script = k.s;
      return true;
    }
  if (in(v, 0xE01, 0xE3A) || in(v, 0xE40, 0xE4E)) { script = kThai; return true; }
  if (in(v, 0xE81, 0xEBD) || in(v, 0xEC0, 0xECD) || in(v, 0xEDC, 0xEDF)) { script = kLao; return true; }
  if (v == 0xF00 || in(v, 0xF40, 0xF6C) || in(v, 0xF71, 0xF84) || in(v, 0xF86, 0xFBC)) { script = kTibetan; return true; }
  if ((in(v, 0x1000, 0x109F) && !in(v, 0x104A, 0x104F) && !in(v, 0x109E, 0x109F)) || in(v, 0xA9E0, 0xA9FE) || in(v, 0xAA60, 0xAA7F)) { script = kMyanmar; return true; }
  if ((in(v, 0x10A0, 0x10FF) && v != 0x10FB) || in(v, 0x1C90, 0x1CBF) || in(v, 0x2D00, 0x2D2D)) { script = kGeorgian; return true; }
  if (in(v, 0x1100, 0x11FF) || in(v, 0x3131, 0x318E) || in(v, 0xA960, 0xA97C) || in(v, 0xAC00, 0xD7A3) || in(v, 0xD7B0, 0xD7FB) ||
      in(v, 0xFFA0, 0xFFDC)) { script = kHangul; return true; }
  if (in(v, 0x1200, 0x135F) || in(v, 0x1380, 0x138F) || in(v, 0x2D80, 0x2DDE)) { script = kEthiopic; return true; }
  if (in(v, 0x1780, 0x17D3) || v == 0x17D7 || v == 0x17DC || v == 0x17DD) { script = kKhmer; return true; }
  if (in(v, 0x1820, 0x1878) || in(v, 0x1880, 0x18AA)) { script = kMongolian; return true; }
  if (in(v, 0x3041, 0x3096) || in(v, 0x3099, 0x309A) || in(v, 0x309D, 0x309F)) { script = kHiragana; return true; }
  if (in(v, 0x30A1, 0x30FA) || in(v, 0x30FC, 0x30FF) || in(v, 0x31F0, 0x31FF) || in(v, 0xFF66, 0xFF9F)) { script = kKatakana; return true; }
  if (in(v, 0x3105, 0x312F) || in(v, 0x31A0, 0x31BF)) { script = kBopomofo; return true; }
  if (in(v, 0x3190, 0x3191) || in(v, 0x3196, 0x319F)) { script = kKanbun; return false; }  // annotation marks: category So, not letters
  if (v == 0x3005 || v == 0x3006 || in(v, 0x3400, 0x4DBF) || in(v, 0x4E00, 0x9FFF) || in(v, 0xF900, 0xFAD9) || in(v, 0x20000, 0x2A6DF) ||
      in(v, 0x2A700, 0x2EBE0) || in(v, 0x2F800, 0x2FA1D) || in(v, 0x30000, 0x3134A)) { script = kHan; return true; }
  if (is_mark(v)) return true;  // a combining mark on its own
  return false;
}

bool is_alphabet_to_segment(int script) {  // src/fuzzy_match.cc:84-90
  return script == kHan || script == kKanbun || script == kKatakana || script == kHiragana || script == kThai || script == kMyanmar;
}

// simple case mapping of the covered cased scripts: 0 = uncased, 1 = lower, 2 = upper; `other` = the counterpart
int letter_case(cp_t v, cp_t& other) {
  other = v;
  if (in(v, 'a', 'z')) { other = v - 32; return 1; }
  if (in(v, 'A', 'Z')) { other = v + 32; return 2; }
  if (v < 0x80) return 0;
  if (v == 0xB5) { other = 0x39C; return 1; }
  if (v == 0xDF || v == 0xAA || v == 0xBA) return 1;  // ß ª º: lower case, no simple upper-case counterpart
  if (in(v, 0xC0, 0xDE) && v != 0xD7) { other = v + 32; return 2; }
  if (in(v, 0xE0, 0xFE) && v != 0xF7) { other = v - 32; return 1; }
  if (v == 0xFF) { other = 0x178; return 1; }
  if (v == 0x178) { other = 0xFF; return 2; }
  if (v == 0x130) { other = 'i'; return 2; }
  if (v == 0x131) { other = 'I'; return 1; }
  if (v == 0x138 || v == 0x149) return 1;
  if (v == 0x17F) { other = 'S'; return 1; }
  auto even_upper = [&](cp_t a, cp_t b) -> int {  // pairs (upper at even code points) inside [a, b]
    if (!in(v, a, b)) return -1;
    if ((v & 1u) == 0) { other = v + 1; return 2; }
    other = v - 1; return 1;
  };
  auto odd_upper = [&](cp_t a, cp_t b) -> int {
    if (!in(v, a, b)) return -1;
    if (v & 1u) { other = v + 1; return 2; }
    other = v - 1; return 1;
  };
  int r;
  if ((r = even_upper(0x100, 0x137)) >= 0) return r;
  if ((r = odd_upper(0x139, 0x148)) >= 0) return r;
  if ((r = even_upper(0x14A, 0x177)) >= 0) return r;
  if ((r = odd_upper(0x179, 0x17E)) >= 0) return r;
  if ((r = even_upper(0x1DE, 0x1EF)) >= 0) return r;
  if ((r = even_upper(0x1F8, 0x21F)) >= 0) return r;
  if ((r = even_upper(0x222, 0x233)) >= 0) return r;
  if ((r = odd_upper(0x1CD, 0x1DC)) >= 0) return r;
  if (v == 0x1DD) { other = 0x18E; return 1; }
  if (in(v, 0x250, 0x2AF)) return 1;  // IPA extensions: lower case
  // Greek
  if (v == 0x386) { other = 0x3AC; return 2; }
  if (in(v, 0x388, 0x38A)) { other = v + 37; return 2; }
  if (v == 0x38C) { other = 0x3CC; return 2; }
  if (v == 0x38E || v == 0x38F) { other = v + 63; return 2; }
  if (in(v, 0x391, 0x3AB) && v != 0x3A2) { other = v + 32; return 2; }
  if (v == 0x3AC) { other = 0x386; return 1; }
  if (in(v, 0x3AD, 0x3AF)) { other = v - 37; return 1; }
  if (v == 0x390 || v == 0x3B0) return 1;
  if (v == 0x3C2) { other = 0x3A3; return 1; }
  if (in(v, 0x3B1, 0x3CB)) { other = v - 32; return 1; }
  if (v == 0x3CC) { other = 0x38C; return 1; }
  if (v == 0x3CD || v == 0x3CE) { other = v - 63; return 1; }
  if ((r = even_upper(0x3D8, 0x3EF)) >= 0) return r;
  // Cyrillic
  if (in(v, 0x400, 0x40F)) { other = v + 80; return 2; }
  if (in(v, 0x410, 0x42F)) { other = v + 32; return 2; }
  if (in(v, 0x430, 0x44F)) { other = v - 32; return 1; }
  if (in(v, 0x450, 0x45F)) { other = v - 80; return 1; }
  if ((r = even_upper(0x460, 0x481)) >= 0) return r;
  if ((r = even_upper(0x48A, 0x4BF)) >= 0) return r;
  if (v == 0x4C0) { other = 0x4CF; return 2; }
  if (v == 0x4CF) { other = 0x4C0; return 1; }
  if ((r = odd_upper(0x4C1, 0x4CE)) >= 0) return r;
  if ((r = even_upper(0x4D0, 0x52F)) >= 0) return r;
  // Armenian
  if (in(v, 0x531, 0x556)) { other = v + 48; return 2; }
  if (in(v, 0x561, 0x586)) { other = v - 48; return 1; }
  if (v == 0x587) return 1;
  // Latin extended additional, Greek extended (the regular part), full-width forms
  if (in(v, 0x1E96, 0x1E9F)) return 1;
  if ((r = even_upper(0x1E00, 0x1EFF)) >= 0) return r;
  if (in(v, 0x1F00, 0x1F6F) && !in(v, 0x1F50, 0x1F5F)) {
    if ((v & 0xF) < 8) { other = v + 8; return 1; }
    other = v - 8; return 2;
  }
  if (in(v, 0x1F70, 0x1FFF)) {
    static const cp_t up[] = {0x1FB8, 0x1FB9, 0x1FBA, 0x1FBB, 0x1FBC, 0x1FC8, 0x1FC9, 0x1FCA, 0x1FCB, 0x1FCC, 0x1FD8, 0x1FD9, 0x1FDA,
                              0x1FDB, 0x1FE8, 0x1FE9, 0x1FEA, 0x1FEB, 0x1FEC, 0x1FF8, 0x1FF9, 0x1FFA, 0x1FFB, 0x1FFC, 0x1F59,
                              0x1F5B, 0x1F5D, 0x1F5F};
    for (cp_t u : up) if (u == v) return 2;
    return 1;
  }
  if (in(v, 0xFF21, 0xFF3A)) { other = v + 32; return 2; }
  if (in(v, 0xFF41, 0xFF5A)) { other = v - 32; return 1; }
  return 0;
}

struct Unit {  // one character with the combining marks that follow it
  std::string s;
  cp_t v;
  bool has_marks;
};

void explode_with_marks(const std::string& text, std::vector<Unit>& units) {
  const unsigned char* p = (const unsigned char*)text.data();
  size_t n = text.size(), i = 0;
  while (i < n) {
    size_t len;
    const cp_t v = decode(p + i, n - i, len);
    if (!units.empty() && is_mark(v)) {
      units.back().s.append((const char*)p + i, len);
      units.back().has_marks = true;
    } else {
      Unit u;
      u.s.assign((const char*)p + i, len);
      u.v = v;
      u.has_marks = false;
      units.push_back(std::move(u));
    }
    i += len;
  }
}

struct Tok {
  std::string s;
  bool jl = false, jr = false;
};

// onmt::CaseModifier: lower-cases the token, returns the case feature
char lowercase_token(std::string& tok) {
  enum { None, Lower, CapFirst, Cap, Upper, Mixed } t = None;
  std::string out;
  const unsigned char* p = (const unsigned char*)tok.data();
  size_t n = tok.size(), i = 0;
  while (i < n) {
    size_t len;
    const cp_t v = decode(p + i, n - i, len);
    cp_t other;
    const int c = letter_case(v, other);
    if (c == 2) {
      append_utf8(out, other);
      t = t == None ? CapFirst : t == CapFirst ? Upper : (t == Lower || t == Cap) ? Mixed : t;
    } else {
      out.append((const char*)p + i, len);
      if (c == 1) t = t == None ? Lower : t == CapFirst ? Cap : t == Upper ? Mixed : t;
    }
    i += len;
  }
  tok.swap(out);
  switch (t) {
    case Lower: return 'L';
    case CapFirst: case Cap: return 'C';
    case Upper: return 'U';
    case Mixed: return 'M';
    default: return 'N';
  }
}

bool is_placeholder(const std::string& tok) { return tok.find(kPhOpen) != std::string::npos; }

}  // namespace

void tokenize_aggressive(int pt, const std::string& text, std::vector<std::string>& tokens, std::vector<std::string>& feats) {
  std::vector<Unit> units;
  explode_with_marks(text, units);
  std::vector<Tok> toks;
  Tok cur;
  bool letter = false, number = false, other = false, space = true, placeholder = false;
  int prev_alphabet = kNoScript;
  auto push_cur = [&](Tok next) {
    toks.push_back(std::move(cur));
    cur = std::move(next);
  };
  for (const Unit& u : units) {
    const cp_t v = u.v;
    const bool sep = is_separator(v) && !u.has_marks;
    if (placeholder) {
      cur.s += u.s;  // NoSubstitution: everything inside a placeholder is kept as it is
      if (v == kPhCloseCp) { letter = true; prev_alphabet = kPlaceholderAlphabet; placeholder = false; space = false; }
    } else if (v == kPhOpenCp) {
      if (!space) {
        Tok next;
        if ((letter && prev_alphabet != kPlaceholderAlphabet) || number) next.jl = true;
        else cur.jr = true;
        push_cur(std::move(next));
      } else if (other && cur.s.empty() && !toks.empty()) {
        toks.back().jr = true;
      }
      cur.s += u.s;
      placeholder = true;
    } else if (sep) {
      if (!space) push_cur(Tok());
      letter = number = other = false;
      space = true;
    } else if (v > 32 && v != 0xFEFF) {
      int alphabet = kNoScript;
      const bool cur_letter = letter_script(v, alphabet);
      const bool cur_number = !cur_letter && is_number(v);
      if (cur_letter) {
        const bool split = (!letter && !space) ||
                           (letter && !is_mark(v) &&
                            ((prev_alphabet == alphabet && is_alphabet_to_segment(alphabet)) || prev_alphabet != alphabet));
        if (split) {
          Tok next;
          next.jl = true;
          push_cur(std::move(next));
        } else if (other && cur.s.empty() && !toks.empty()) {
          toks.back().jr = true;
        }
        cur.s += u.s;
        letter = true; number = false; other = false; space = false;
        prev_alphabet = alphabet;
      } else if (cur_number) {
        if (letter || (!number && !space)) {
          Tok next;
          if (!letter || prev_alphabet == kPlaceholderAlphabet) cur.jr = true;
          else next.jl = true;
          push_cur(std::move(next));
        } else if (other && cur.s.empty() && !toks.empty()) {
          toks.back().jr = true;
        }
        cur.s += u.s;
        letter = false; number = true; other = false; space = false;
      } else {  // any other character is a token of its own
        if (!space) {
          Tok next;
          next.jl = true;
          push_cur(std::move(next));
        } else if (other) {
          cur.jl = true;
        }
        cur.s += u.s;
        push_cur(Tok());
        letter = number = false;
        other = true;
        space = true;
      }
    }
  }
  if (!cur.s.empty()) toks.push_back(std::move(cur));
  // finalize: case features, joiner / spacer markers as tokens of their own
  const bool cas = (pt & kPtCas) != 0, jnr = (pt & kPtJnr) != 0, spc = !jnr && (pt & kPtSep) != 0;
  tokens.clear();
  feats.clear();
  auto emit = [&](const std::string& s, char f) {
    tokens.push_back(s);
    feats.push_back(std::string(1, f));
  };
  bool last_was_joiner = false;
  for (size_t i = 0; i < toks.size(); i++) {
    Tok& t = toks[i];
    if (t.s.empty()) continue;
    const bool joined_left = t.jl || (i > 0 && toks[i - 1].jr);
    if (jnr && joined_left && i > 0 && !last_was_joiner) emit(kJoiner, 'N');
    if (spc && !joined_left && i > 0) emit(kSpacer, 'N');
    char f = 'N';
    if (cas && !is_placeholder(t.s)) f = lowercase_token(t.s);
    emit(t.s, f);
    last_was_joiner = false;
    if (jnr && t.jr && i + 1 < toks.size()) { emit(kJoiner, 'N'); last_was_joiner = true; }
  }
}

// The same function for a sentence of ASCII bytes under penalty tokens that emit no joiner / spacer tokens (round 5: the ingestion
// path, fmx_index_add_text_bulk, spent most of its time building a std::string per character).  On ASCII the tokenizer above is a
// byte-class machine — ASCII is NFC, every letter is Latin and cased by one bit, there are no placeholders and no combining marks:
// a run of letters / of digits is a token, any other printable byte a token of its own, white space separates, bytes below 33 that
// are not white space are dropped without ending a token — and _tokenize_and_normalize keeps a word's case feature as its real
// token (cas), turns a number into the placeholder (nbr) and punctuation into a penalty string (pct).  Same output as the general
// path: tests/test_text_frontend.py::test_ascii_fast_path_equals_the_general_tokenizer.
void tokenize_and_normalize_general(int pt, const std::string& sentence, TextTokens& out);

// the machine: emit(token — lower-cased when it is a word under cas —, class: 0 other / 1 letters / 2 digits, case feature)
template <class Emit>
static bool ascii_machine(int pt, const unsigned char* p, size_t n, Emit&& emit) {
  if (pt & (kPtJnr | kPtSep)) return false;
  for (size_t k = 0; k < n; k++) if (p[k] >= 0x80) return false;
  const bool cas = (pt & kPtCas) != 0;
  enum { kNone = 0, kLetter = 1, kDigit = 2 };
  int cur = kNone;
  std::string tok;
  auto flush = [&]() {
    if (tok.empty()) { cur = kNone; return; }
    char f = 'N';
    if (cas && cur == kLetter) {  // onmt::CaseModifier: lower case + case feature
      size_t n_up = 0;
      const bool first_up = tok[0] >= 'A' && tok[0] <= 'Z';
      for (char& ch : tok) if (ch >= 'A' && ch <= 'Z') { ch = (char)(ch | 0x20); n_up++; }
      f = n_up == 0 ? 'L' : ((first_up && n_up == 1) ? 'C' : ((n_up == tok.size() && tok.size() >= 2) ? 'U' : 'M'));
    }
    emit(tok, cur, f);
    tok.clear();
    cur = kNone;
  };
  for (size_t k = 0; k < n; k++) {
    const unsigned char c = p[k];
    const unsigned char l = c | 0x20;
    if (l >= 'a' && l <= 'z') { if (cur != kLetter) { flush(); cur = kLetter; } tok += (char)c; }
    else if (c >= '0' && c <= '9') { if (cur != kDigit) { flush(); cur = kDigit; } tok += (char)c; }
    else if (c == ' ' || (c >= 9 && c <= 13)) flush();
    else if (c > 32) { flush(); tok += (char)c; cur = kNone; flush(); }
    // (other bytes below 33: dropped, the token at hand goes on)
  }
  flush();
  return true;
}

static const std::string& num_placeholder() {
  static const std::string k = std::string(kPhOpen) + "num" + kPhClose;
  return k;
}

static bool tokenize_and_normalize_ascii(int pt, const std::string& sentence, TextTokens& out) {
  const bool cas = (pt & kPtCas) != 0, nbr = (pt & kPtNbr) != 0, pct = (pt & kPtPct) != 0;
  out.map_tokens.push_back(0);
  return ascii_machine(pt, (const unsigned char*)sentence.data(), sentence.size(), [&](const std::string& tok, int cls, char f) {
    const size_t i = out.raw.size();
    out.raw.push_back(tok);
    out.feat.push_back(std::string(1, f));
    const uint64_t real_i = out.norm.size();
    auto push = [&](const std::string& norm, const std::string& real) {
      out.norm.push_back(norm); out.real.push_back(real); out.map_tokens.push_back((uint32_t)i + 1);
    };
    if (cas && f != 'N') push(tok, std::string(1, f));
    else if (cls == 2) { if (nbr) push(num_placeholder(), tok); else push(tok, tok); }
    else if (cls == 0 && pct) {  // punctuation: a penalty string in front of the next real token (appended to one already there)
      if (!out.itoks.empty() && out.itoks.back().first == real_i) out.itoks.back().second += tok;
      else out.itoks.emplace_back(real_i, tok);
    } else push(tok, tok);
  });
}

// The same into packed storage (the bulk ingestion path: no string per token): the normalised tokens NUL-terminated behind one
// another appended to `norm`, the real tokens tab-joined appended to `real` — what Sentence::_tokstring holds (include/fuzzy/sentence.hh).
bool tokenize_and_normalize_packed(int pt, const char* sentence, size_t len, std::string& norm, uint32_t& n_tok, std::string& real,
                                   std::vector<std::pair<uint64_t, std::string>>& itoks) {
  const bool cas = (pt & kPtCas) != 0, nbr = (pt & kPtNbr) != 0, pct = (pt & kPtPct) != 0;
  const size_t norm0 = norm.size(), real0 = real.size();
  n_tok = 0;
  itoks.clear();
  auto push = [&](const std::string& nm, const char* r, size_t rl) {
    norm.append(nm); norm.push_back('\0');
    if (n_tok) real.push_back('\t');
    real.append(r, rl);
    n_tok++;
  };
  const bool ok = ascii_machine(pt, (const unsigned char*)sentence, len, [&](const std::string& tok, int cls, char f) {
    if (cas && f != 'N') push(tok, &f, 1);
    else if (cls == 2) { if (nbr) push(num_placeholder(), tok.data(), tok.size()); else push(tok, tok.data(), tok.size()); }
    else if (cls == 0 && pct) {
      if (!itoks.empty() && itoks.back().first == n_tok) itoks.back().second += tok;
      else itoks.emplace_back((uint64_t)n_tok, tok);
    } else push(tok, tok.data(), tok.size());
  });
  if (ok) return true;
  norm.resize(norm0); real.resize(real0);
  TextTokens t;
  tokenize_and_normalize_general(pt, std::string(sentence, len), t);
  n_tok = (uint32_t)t.norm.size();
  for (size_t k = 0; k < t.norm.size(); k++) {
    norm.append(t.norm[k]); norm.push_back('\0');
    if (k) real.push_back('\t');
    real.append(t.real[k]);
  }
  itoks = std::move(t.itoks);
  return true;
}

// the general path alone (tests compare the ASCII fast path with it)
void tokenize_and_normalize_general(int pt, const std::string& sentence, TextTokens& out);

// src/fuzzy_match.cc:108-193
void tokenize_and_normalize(int pt, const std::string& sentence, TextTokens& out) {
  out = TextTokens();
  if (tokenize_and_normalize_ascii(pt, sentence, out)) return;
  out = TextTokens();
  tokenize_and_normalize_general(pt, sentence, out);
}

void tokenize_and_normalize_general(int pt, const std::string& sentence, TextTokens& out) {
  out = TextTokens();
  tokenize_aggressive(pt, nfc_normalize(sentence), out.raw, out.feat);  // src/fuzzy_match.cc:115-116
  out.map_tokens.push_back(0);
  auto set_itok = [&](uint64_t idx, const std::string& s) {  // Sentence::set_itok appends (include/fuzzy/sentence.hxx)
    if (!out.itoks.empty() && out.itoks.back().first == idx) out.itoks.back().second += s;
    else out.itoks.emplace_back(idx, s);
  };
  auto push = [&](const std::string& norm, const std::string& real, size_t i) {
    out.norm.push_back(norm);
    out.real.push_back(real);
    out.map_tokens.push_back((uint32_t)i + 1);
  };
  const std::string open(kPhOpen), close(kPhClose);
  for (size_t i = 0; i < out.raw.size(); i++) {
    const std::string& token = out.raw[i];
    const uint64_t real_i = out.norm.size();
    if (token == kSpacer || token == kJoiner) { set_itok(real_i, " "); continue; }
    if ((pt & kPtCas) && out.feat[i] != "N") { push(token, out.feat[i], i); continue; }  // word tokens keep only the case feature
    if (is_placeholder(token)) {
      const size_t ph_begin = token.find(open);
      size_t ph_end = token.find("\xEF\xBC\x83", ph_begin);                       // ＃ id separator
      if (ph_end == std::string::npos) {
        ph_end = token.find("\xEF\xBC\x9A", ph_begin);                            // ： value separator
        if (ph_end == std::string::npos) {
          ph_end = token.find(close, ph_begin);
          if (ph_end == std::string::npos) ph_end = token.length();
        }
      }
      std::string ent = token.substr(ph_begin + open.size(), ph_end - ph_begin - open.size());
      if (ent.length() >= 2 && ent.substr(0, 2) == "it") ent = "it";
      if (ent == "it" && (pt & kPtTag)) set_itok(real_i, "T");
      else push(open + ent + close, token, i);
      continue;
    }
    size_t len;
    const cp_t cp = decode((const unsigned char*)token.data(), token.size(), len);
    int script;
    if (is_number(cp)) {
      if (pt & kPtNbr) push(open + "num" + close, token, i);
      else push(token, token, i);
    } else if (!letter_script(cp, script) && (pt & kPtPct)) {
      set_itok(real_i, token);
    } else {
      push(token, token, i);
    }
  }
}

std::string detokenize(int pt, const std::vector<std::string>& tokens, const std::vector<std::string>& feats) {
  const bool jnr = (pt & kPtJnr) != 0, spc = !jnr && (pt & kPtSep) != 0;
  std::string out;
  bool glue = true;  // no space in front of the next token
  for (size_t i = 0; i < tokens.size(); i++) {
    const std::string& t = tokens[i];
    if (jnr && t == kJoiner) { glue = true; continue; }
    if (spc && t == kSpacer) { out += ' '; glue = true; continue; }
    std::string w = t;
    const char f = i < feats.size() && !feats[i].empty() ? feats[i][0] : 'N';
    if (f == 'C' || f == 'U') {  // restore the casing the feature records
      std::string r;
      const unsigned char* p = (const unsigned char*)w.data();
      size_t n = w.size(), k = 0;
      bool first = true;
      while (k < n) {
        size_t len;
        const cp_t v = decode(p + k, n - k, len);
        cp_t other;
        if (letter_case(v, other) == 1 && other != v && (f == 'U' || first)) append_utf8(r, other);
        else r.append((const char*)p + k, len);
        first = false;
        k += len;
      }
      w.swap(r);
    }
    if (!glue && !spc) out += ' ';
    out += w;
    glue = false;
  }
  return out;
}

}  // namespace fmx
