/* Embeds the mainnet trusted setup (public ceremony data; the reference embeds the same data as JSON,
   crates/trusted_setup/src/lib.rs:5,23-35) into the shared library. SRS_PATH is set by the Makefile. */
    .section .rodata
    .global kzg_srs_begin
    .global kzg_srs_end
    .balign 16
kzg_srs_begin:
    .incbin SRS_PATH
kzg_srs_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
