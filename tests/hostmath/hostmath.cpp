// Test-only host build of the device math headers (eigensolver_amd/csrc/es_bessel.hpp) so that the Bessel
// routines can be checked against scipy.special on the CPU.  Not part of the product library.
#define ES_HD static inline
#include "../../eigensolver_amd/csrc/es_bessel.hpp"
extern "C" {
void hm_ke_pair(int n, double x, double* out) { esb::ke_pair(n, x, out[0], out[1]); }
void hm_ie_pair(int n, double x, double* out) { esb::ie_pair(n, x, out[0], out[1]); }
void hm_ie_pair_from_k(int n, double x, double* out) {
  double kn, kn1;
  esb::ke_pair(n, x, kn, kn1);
  esb::ie_pair_from_k(n, x, kn, kn1, out[0], out[1]);
}
void hm_jy_pair(int n, double x, double* out) { esb::jy_pair(n, x, out[0], out[1], out[2], out[3]); }
}
