/* jit_embed.S — the text of the device library, as data: hiprtc compiles it at run time together with each generated
 * per-circuit kernel (jit_engine.hip).  These are the very files hipcc compiles into the ahead-of-time kernels. */
    .section .rodata
#define EMBED(sym, file) \
    .global sym; \
sym: \
    .incbin file; \
    .byte 0
EMBED(dusp_src_device_types, "device_types.hpp")
EMBED(dusp_src_device_util, "device_util.hpp")
EMBED(dusp_src_filter_lamda, "filter_lamda.hpp")
EMBED(dusp_src_map_ops, "map_ops.hpp")
EMBED(dusp_src_repeat_add, "repeat_add.hpp")
EMBED(dusp_src_jit_args, "jit_args.hpp")
EMBED(dusp_src_jit_prelude, "jit_prelude.hpp")
    .section .note.GNU-stack,"",@progbits
