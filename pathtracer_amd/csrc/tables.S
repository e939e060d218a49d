/* Embeds the blobs of pathtracer_amd/data into libptcore.so (assembled with -I ../data). */
    .section .rodata
    .macro blob sym, file
    .balign 16
    .globl \sym
    .hidden \sym
    .type \sym, @object
\sym:
    .incbin "\file"
    .size \sym, . - \sym
    .endm
    blob ptc_tab_hw_datasets,     "hw_datasets.bin"
    blob ptc_tab_hw_datasets_rad, "hw_datasets_rad.bin"
    blob ptc_tab_hw_solar,        "hw_solar.bin"
    blob ptc_tab_hw_limb,         "hw_limb.bin"
    blob ptc_tab_srgb2refl,       "srgb2refl.bin"
    blob ptc_tab_d65,             "d65.bin"
    .section .note.GNU-stack,"",@progbits
