#include <cstdint>
extern "C" uint64_t ldp_phased_phase_offset(uint32_t hap_ct) { return (((uint64_t)(hap_ct / 2) + 3) / 4 + 3) & ~(uint64_t)3; }
extern "C" uint64_t ldp_phased_row_bytes(uint32_t hap_ct) { return ldp_phased_phase_offset(hap_ct) + ((uint64_t)(hap_ct / 2) + 7) / 8; }
