/* Embeds the committed data blobs (the blobs of pathtracer_amd/data, see tools/gen_tables.py) into the
 * oracle library.  Assembled with -I pathtracer_amd/data. */
    .section .rodata
    .macro blob sym, file
    .balign 16
    .globl \sym
    .type \sym, @object
\sym:
    .incbin "\file"
    .size \sym, . - \sym
    .endm
    blob pto_tab_hw_datasets,     "hw_datasets.bin"
    blob pto_tab_hw_datasets_rad, "hw_datasets_rad.bin"
    blob pto_tab_hw_solar,        "hw_solar.bin"
    blob pto_tab_hw_limb,         "hw_limb.bin"
    blob pto_tab_srgb2refl,       "srgb2refl.bin"
    blob pto_tab_d65,             "d65.bin"
    .section .note.GNU-stack,"",@progbits
