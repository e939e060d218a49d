// tables.h — data blobs embedded by tables.S (pathtracer_amd/data/*.bin; tools/gen_tables.py).
#ifndef PTC_TABLES_H
#define PTC_TABLES_H
extern "C" {
extern const float ptc_tab_hw_datasets[];      // [11][2][10][6][9]   sky-dome coefficients
extern const float ptc_tab_hw_datasets_rad[];  // [11][2][10][6]      mean radiances
extern const float ptc_tab_hw_solar[];         // [11][10][45][4]     solar-disc piecewise cubics
extern const float ptc_tab_hw_limb[];          // [11][6]             limb darkening
extern const float ptc_tab_srgb2refl[];        // [471][3]            sRGB -> reflectance basis, 360..830 nm
extern const float ptc_tab_d65[];              // [531]               D65, 300..830 nm
}
#endif
