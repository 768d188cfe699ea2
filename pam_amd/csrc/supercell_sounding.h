// supercell_sounding.h -- the analytic sounding behind both idealised supercell initialisations of the reference:
// Dycore::init_supercell (dynamics/awfl/Dycore.h:777-830 helper functions, :1096-1230 column integration) and the standalone
// driver's supercell_init (standalone/mmf_simplified/supercell_init.h:7-135 with pam_core/idealized_profiles.h).
// Host and device (the column integration of Dycore::init runs on the host once; the driver's column is a kernel).
//
// One object holds the sounding: the piecewise-linear temperature profile (300 K at the ground, 213 K from the 12 km
// tropopause up), with the lapse rates, the hydrostatic exponents g/(R_d lapse) and the tropopause pressure formed once --
// each with exactly the expression the reference evaluates at every call, so the values are the same.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PAMA_HD __host__ __device__ inline
#else
#define PAMA_HD inline
#endif

namespace pama {

struct Sounding {
  double z_0, z_trop, z_top, T_0, T_trop, T_top, p_0, R_d, grav;
  double lapse_lo, lapse_hi;   // -(dT/dz) below / above the tropopause
  double expo_lo, expo_hi;     // g / (R_d lapse)
  double p_trop;               // dry pressure at the tropopause

  PAMA_HD static Sounding make(double z_top, double R_d, double grav) {
    Sounding s;
    s.z_0 = 0; s.z_trop = 12000; s.z_top = z_top; s.T_0 = 300; s.T_trop = 213; s.T_top = 213; s.p_0 = 100000;
    s.R_d = R_d; s.grav = grav;
    s.lapse_lo = -(s.T_trop - s.T_0) / (s.z_trop - s.z_0);
    s.lapse_hi = -(s.T_top - s.T_trop) / (s.z_top - s.z_trop);
    s.expo_lo = grav / (R_d * s.lapse_lo);
    s.expo_hi = grav / (R_d * s.lapse_hi);       // unused (inf) for the isothermal stratosphere
    s.p_trop = s.p_0 * pow(s.T_trop / s.T_0, s.expo_lo);
    return s;
  }
  PAMA_HD double temperature(double z) const {
    return z <= z_trop ? T_0 - lapse_lo * (z - z_0) : T_trop - lapse_hi * (z - z_trop);
  }
  PAMA_HD double pressure_dry(double z) const {
    if (z <= z_trop) return p_0 * pow(temperature(z) / T_0, expo_lo);
    if (lapse_hi != 0) return p_trop * pow(temperature(z) / T_trop, expo_hi);
    return p_trop * exp(-grav * (z - z_trop) / (R_d * T_trop));
  }
  PAMA_HD double relhum(double z) const { return z <= z_trop ? 1.0 - 0.75 * pow(z / z_trop, 1.25) : 0.25; }
  PAMA_HD static double sat_mix_dry(double press, double T) { return 380 / (press)*exp(17.27 * (T - 273) / (T - 36)); }
  // water-vapour mixing ratio at height z, capped at 0.014 (Dycore.h:1147-1152, supercell_init.h:59-63); also the temperature.
  // A quadrature point can sit at z = -1e-13, where pow() returns NaN: the reference's min(0.014, NaN) is 0.014 -- fmin().
  PAMA_HD double vapour_mixing_ratio(double z, double &temp) const {
    temp = temperature(z);
    const double qvs = sat_mix_dry(pressure_dry(z), temp);
    double rh = relhum(z);
    if (rh * qvs > 0.014) rh = 0.014 / qvs;
    return fmin(0.014, qvs * rh);
  }
};

}  // namespace pama
