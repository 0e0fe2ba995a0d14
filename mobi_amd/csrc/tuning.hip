#include <atomic>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "tuning.h"

namespace mobi {
namespace {
Tuning g_tuning;
std::once_flag g_once;

int env_int(const char* name) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : -1;
}
void read_env() {
  g_tuning.persist_blocks = env_int("MOBI_IGEMM_PERSIST_BLOCKS");
  g_tuning.pp_split = env_int("MOBI_IGEMM_PP_SPLIT");
  g_tuning.wm = env_int("MOBI_IGEMM_WM");
  g_tuning.fast = env_int("MOBI_IGEMM_FAST");
  g_tuning.glds = env_int("MOBI_IGEMM_GLDS");
  g_tuning.lin = env_int("MOBI_IGEMM_LIN");
  g_tuning.epi_direct = env_int("MOBI_IGEMM_EPI_DIRECT");
  g_tuning.pp = env_int("MOBI_IGEMM_PP");
  g_tuning.sm = env_int("MOBI_IGEMM_SM");
  g_tuning.wide = env_int("MOBI_IGEMM_WIDE");
  g_tuning.wide128 = env_int("MOBI_IGEMM_WIDE128");
  g_tuning.ring_direct = env_int("MOBI_IGEMM_RING_DIRECT");
  g_tuning.sm_direct = env_int("MOBI_IGEMM_SM_DIRECT");
  g_tuning.w_tiled = env_int("MOBI_IGEMM_WTILED");
  g_tuning.sm64 = env_int("MOBI_IGEMM_SM64");
  g_tuning.n_major = env_int("MOBI_IGEMM_N_MAJOR");
  g_tuning.split_target = env_int("MOBI_IGEMM_SPLIT_TARGET");
  g_tuning.split_longk = env_int("MOBI_IGEMM_SPLIT_LONGK");
  g_tuning.split_round4 = env_int("MOBI_IGEMM_SPLIT_ROUND4");
  g_tuning.attn_bwd_exact_d = env_int("MOBI_ATTN_BWD_EXACT_D");
  g_tuning.fused_split = env_int("MOBI_IGEMM_FUSED_SPLIT");
  g_tuning.small = env_int("MOBI_IGEMM_SMALL");
  g_tuning.small_mflop = env_int("MOBI_IGEMM_SMALL_MFLOP");
  g_tuning.small_conv_m = env_int("MOBI_IGEMM_SMALL_CONV_M");
  g_tuning.tka_mfma = env_int("MOBI_TKA_MFMA");
  g_tuning.attn_nw = env_int("MOBI_ATTN_NW");
  g_tuning.attn_sp = env_int("MOBI_ATTN_SP");
  g_tuning.attn_v3 = env_int("MOBI_ATTN_V3");
  g_tuning.gn_fused = env_int("MOBI_GN_FUSED");
  g_tuning.gn_coop = env_int("MOBI_GN_COOP");
  g_tuning.gn_split_pw4 = env_int("MOBI_GN_SPLIT_PW4");
  g_tuning.skinny_mfma = env_int("MOBI_SKINNY_MFMA");
  g_tuning.cout_mfma = env_int("MOBI_COUT_MFMA");
  g_tuning.attn_xcd = env_int("MOBI_ATTN_XCD");
  g_tuning.attn_h16 = env_int("MOBI_ATTN_H16");
  g_tuning.attn_nw8_blocks = env_int("MOBI_ATTN_NW8_BLOCKS");
  g_tuning.tka_rows = env_int("MOBI_TKA_ROWS");
}
}  // namespace

const Tuning& tuning() {
  std::call_once(g_once, read_env);
  return g_tuning;
}
}  // namespace mobi

extern "C" int mobi_tuning_reload(void) {
  std::call_once(mobi::g_once, [] {});
  mobi::read_env();
  return MOBI_OK;
}

extern "C" int mobi_build_info(void) {
#ifdef MOBI_DEV
  return 1;
#else
  return 0;
#endif
}
