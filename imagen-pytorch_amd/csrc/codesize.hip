// codesize.hip — device code size of a kernel of THIS library, by symbol name (host code only).
//
// Why: inside the denoiser step every launch runs a different kernel from its predecessor and the 20-70 KB of an igemm instantiation
// were last executed a whole step (GBs of traffic) ago, so a launch starts with a chain of instruction-cache misses served from HBM
// (round-2 probe latency_probe.py --separate: +3 us per launch when the code is still in L2, several times that from HBM).  The kernels
// therefore read their own code range as DATA once at start (one parallel round trip that fills the XCD's L2), which needs the
// size of the function.  The HIP runtime has no query for it, so the symbol tables of the gfx950 code objects embedded in this shared
// object (section .hip_fatbin: clang offload bundles, one per translation unit) are read once from the file on disk.
#include <dlfcn.h>
#include <elf.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace {

std::map<std::string, unsigned> g_sizes;
std::once_flag g_once;

void scan_code_object(const unsigned char* co, size_t len) {
  if (len < sizeof(Elf64_Ehdr) || memcmp(co, ELFMAG, SELFMAG) != 0 || co[EI_CLASS] != ELFCLASS64) return;
  const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(co);
  if (eh->e_shoff == 0 || eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) > len) return;
  const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(co + eh->e_shoff);
  for (int i = 0; i < eh->e_shnum; ++i) {
    if (sh[i].sh_type != SHT_SYMTAB || sh[i].sh_link >= eh->e_shnum) continue;
    const Elf64_Shdr& st = sh[sh[i].sh_link];
    if (sh[i].sh_offset + sh[i].sh_size > len || st.sh_offset + st.sh_size > len) continue;
    const Elf64_Sym* sym = reinterpret_cast<const Elf64_Sym*>(co + sh[i].sh_offset);
    const char* str = reinterpret_cast<const char*>(co + st.sh_offset);
    const size_t n = sh[i].sh_size / sizeof(Elf64_Sym);
    for (size_t k = 0; k < n; ++k)
      if (ELF64_ST_TYPE(sym[k].st_info) == STT_FUNC && sym[k].st_size > 0 && sym[k].st_name < st.sh_size)
        g_sizes[str + sym[k].st_name] = (unsigned)sym[k].st_size;
  }
}

void load_sizes() {
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void*>(&imagen_set_error), &info) || !info.dli_fname) return;
  FILE* f = fopen(info.dli_fname, "rb");
  if (!f) return;
  std::vector<unsigned char> buf;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz > 0) {
    buf.resize((size_t)sz);
    if (fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear();
  }
  fclose(f);
  static const char kMagic[] = "__CLANG_OFFLOAD_BUNDLE__";
  const size_t ml = sizeof(kMagic) - 1;
  for (size_t pos = 0; pos + ml + 8 <= buf.size(); ++pos) {
    if (buf[pos] != '_' || memcmp(buf.data() + pos, kMagic, ml) != 0) continue;
    const unsigned char* b = buf.data() + pos;
    const size_t room = buf.size() - pos;
    uint64_t nb;
    memcpy(&nb, b + ml, 8);
    size_t cur = ml + 8;
    for (uint64_t e = 0; e < nb && e < 64; ++e) {
      if (cur + 24 > room) break;
      uint64_t off, len, tl;
      memcpy(&off, b + cur, 8);
      memcpy(&len, b + cur + 8, 8);
      memcpy(&tl, b + cur + 16, 8);
      cur += 24;
      if (cur + tl > room) break;
      const std::string triple(reinterpret_cast<const char*>(b + cur), (size_t)tl);
      cur += tl;
      if (triple.find("gfx950") != std::string::npos && off + len <= room) scan_code_object(b + off, (size_t)len);
    }
    pos += ml;
  }
}

}  // namespace

// bytes of device code of the kernel with this (mangled) symbol name; 0 when unknown (the kernels then skip the warm-up)
unsigned imagen_kernel_code_bytes(const char* mangled) {
  std::call_once(g_once, load_sizes);
  auto it = g_sizes.find(mangled);
  return it == g_sizes.end() ? 0u : it->second;
}

extern "C" int imagen_debug_code_bytes(const char* mangled) { return (int)imagen_kernel_code_bytes(mangled); }
extern "C" int imagen_debug_num_kernels(void) {
  std::call_once(g_once, load_sizes);
  return (int)g_sizes.size();
}
