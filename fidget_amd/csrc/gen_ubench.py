"""fh_ubench - instruction-cost micro-benchmarks for the interpreters (diagnostics; tools/ubench.py).

kernarg { float* out; u32 test; u32 iters }.  Every wave runs `iters` times a body of 32 copies of the pattern of test
`test` between two s_memtime and stores (clocks / (iters * 32)) as f32 at out[workgroup].  128 VGPRs are allocated so that
at most 4 waves share a SIMD, as in fh_tiles_v32 / fh_columns.
"""

TESTS = [
    ("v_mov independent", lambda k: f"\tv_mov_b32 v{10 + k % 16}, v{40 + k % 8}"),
    ("v_add_f32 dependent chain", lambda k: "\tv_add_f32 v10, v10, v11"),
    ("s_add_u32 dependent chain", lambda k: "\ts_add_u32 s20, s20, 1"),
    ("s_branch taken (to the next instruction)", lambda k: f"\ts_branch .Lub_t3_{k}\n.Lub_t3_{k}:"),
    ("s_add + s_setpc to the next instruction", lambda k: f"\ts_add_u32 s30, s30, 8\n\ts_setpc_b64 s[30:31]"),
    ("v_readlane independent", lambda k: f"\tv_readlane_b32 s{40 + k % 8}, v{40 + k % 8}, s21"),
    ("s_set_gpr_idx_on + relative v_mov", lambda k: f"\ts_set_gpr_idx_on s22, 1\n\tv_mov_b32 v{10 + k % 16}, v40"),
    ("v_cndmask with an SGPR mask", lambda k: f"\tv_cndmask_b32_e64 v{10 + k % 16}, v40, v41, s[24:25]"),
    ("v_pk_add_f32", lambda k: f"\tv_pk_add_f32 v[{10 + 2 * (k % 8)}:{11 + 2 * (k % 8)}], v[40:41], v[42:43]"),
    ("v_mov ; s_add alternating (independent)", lambda k: f"\tv_mov_b32 v{10 + k % 16}, v40\n\ts_add_u32 s{40 + k % 8}, s21, 1"),
    ("s_cmp + s_cbranch_scc1 not taken", lambda k: "\ts_cmp_eq_u32 s20, 1\n\ts_cbranch_scc1 .Lub_exit"),
    ("s_nop 0", lambda k: "\ts_nop 0"),
    ("v_pk_mov_b32", lambda k: f"\tv_pk_mov_b32 v[{10 + 2 * (k % 8)}:{11 + 2 * (k % 8)}], v[40:41], v[40:41] op_sel:[0,1]"),
    ("v_cmp_lt_f32 to SGPR pair", lambda k: f"\tv_cmp_lt_f32_e64 s[{40 + 2 * (k % 4)}:{41 + 2 * (k % 4)}], v40, v41"),
    ("dispatch: 4 v_readlane + s_add + idx_on + v_pk_mov + s_setpc (next instruction)",
     lambda k: "\ts_set_gpr_idx_off\n\tv_readlane_b32 s40, v40, s21\n\tv_readlane_b32 s41, v41, s21\n\tv_readlane_b32 s42, v42, s21\n\tv_readlane_b32 s43, v43, s21\n"
               "\ts_add_u32 s30, s30, 56\n\ts_set_gpr_idx_on s22, 3\n\tv_pk_mov_b32 v[10:11], v[44:45], v[44:45] op_sel:[0,1]\n\ts_setpc_b64 s[30:31]"),
    ("ds_read_b32 dependent (address from the value)", lambda k: "\tds_read_b32 v10, v10\n\ts_waitcnt lgkmcnt(0)"),
    ("v_mul_f32 x4 dependent + s_branch taken", lambda k: f"\tv_mul_f32 v10, v10, v11\n\tv_mul_f32 v10, v10, v11\n\tv_mul_f32 v10, v10, v11\n\tv_mul_f32 v10, v10, v11\n\ts_branch .Lub_t16_{k}\n.Lub_t16_{k}:"),
    ("s_mov_b64 exec + v_mov", lambda k: f"\ts_mov_b64 exec, s[24:25]\n\tv_mov_b32 v{10 + k % 16}, v40"),
    # ---- round 6: what an op of the leaf interpreter costs, piece by piece
    ("v_add_f32 independent", lambda k: f"\tv_add_f32 v{10 + k % 16}, v40, v41"),
    ("v_min_f32 independent", lambda k: f"\tv_min_f32 v{10 + k % 16}, v40, v41"),
    ("v_cmp_lt_f32 vcc (e32) + v_cndmask (vcc)", lambda k: f"\tv_cmp_lt_f32 vcc, v40, v41\n\tv_cndmask_b32 v{10 + k % 16}, v40, v41, vcc"),
    ("v_readfirstlane independent", lambda k: f"\tv_readfirstlane_b32 s{40 + k % 8}, v{40 + k % 6}"),
    ("op as built: idx_off, 3 v_readlane, s_add, s_lshr, s_setpc; idx_on, 4 v_pk_add in place (relative), s_setpc",
     lambda k: "\ts_set_gpr_idx_off\n\tv_readlane_b32 s40, v40, s21\n\tv_readlane_b32 s41, v41, s21\n\tv_readlane_b32 s42, v42, s21\n"
               "\ts_add_u32 s30, s30, 40\n\ts_lshr_b32 s43, s41, 8\n\ts_setpc_b64 s[30:31]\n"
               "\ts_set_gpr_idx_on s22, 9\n\tv_pk_add_f32 v[10:11], v[10:11], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[12:13], v[12:13], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n"
               "\tv_pk_add_f32 v[14:15], v[14:15], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[16:17], v[16:17], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n"
               "\ts_add_u32 s30, s30, 44\n\ts_setpc_b64 s[30:31]"),
    ("op with fixed registers: 2 v_readlane, s_add, s_setpc; 4 v_pk_add (no index mode), s_setpc",
     lambda k: "\tv_readlane_b32 s40, v40, s21\n\tv_readlane_b32 s42, v42, s21\n"
               "\ts_add_u32 s30, s30, 24\n\ts_setpc_b64 s[30:31]\n"
               "\tv_pk_add_f32 v[10:11], v[10:11], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[12:13], v[12:13], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n"
               "\tv_pk_add_f32 v[14:15], v[14:15], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[16:17], v[16:17], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n"
               "\ts_add_u32 s30, s30, 40\n\ts_setpc_b64 s[30:31]"),
    ("the work alone: 4 v_pk_add with an SGPR pair", lambda k: "\tv_pk_add_f32 v[10:11], v[10:11], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[12:13], v[12:13], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n"
               "\tv_pk_add_f32 v[14:15], v[14:15], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]\n\tv_pk_add_f32 v[16:17], v[16:17], s[42:43] op_sel:[0,1] op_sel_hi:[1,1]"),
    ("the work alone, unpacked: 8 v_add_f32 with an SGPR", lambda k: "\n".join(f"\tv_add_f32 v{10 + j}, s43, v{10 + j}" for j in range(8))),
    ("the dispatch alone: idx_off, 3 v_readlane, s_add, s_lshr, s_setpc", 
     lambda k: "\ts_set_gpr_idx_off\n\tv_readlane_b32 s40, v40, s21\n\tv_readlane_b32 s41, v41, s21\n\tv_readlane_b32 s42, v42, s21\n"
               "\ts_add_u32 s30, s30, 40\n\ts_lshr_b32 s43, s41, 8\n\ts_setpc_b64 s[30:31]"),
    ("min in place as built: 4 v_pk_mov, nan test (3 v_pk_add, v_add, v_cmp, branch), 8 v_cmp to SGPRs + 8 v_cndmask",
     lambda k: "\n".join([f"\tv_pk_mov_b32 v[{26 + 2 * j}:{27 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}] op_sel:[0,1]" for j in range(4)] +
                         ["\tv_pk_add_f32 v[34:35], v[10:11], v[12:13]", "\tv_pk_add_f32 v[36:37], v[14:15], v[16:17]", "\tv_pk_add_f32 v[34:35], v[34:35], v[36:37]", "\tv_add_f32 v34, v34, v35",
                          "\tv_cmp_u_f32 vcc, v34, v34", "\ts_cbranch_vccnz .Lub_exit"] +
                         [f"\tv_cmp_nlt_f32_e64 s[{40 + 2 * (j % 4)}:{41 + 2 * (j % 4)}], v{10 + j}, v{26 + j}" for j in range(4)] +
                         [f"\tv_cndmask_b32_e64 v{10 + j}, v{10 + j}, v{26 + j}, s[{40 + 2 * (j % 4)}:{41 + 2 * (j % 4)}]" for j in range(4)] +
                         [f"\tv_cmp_nlt_f32_e64 s[{40 + 2 * (j % 4)}:{41 + 2 * (j % 4)}], v{10 + j}, v{26 + j}" for j in range(4, 8)] +
                         [f"\tv_cndmask_b32_e64 v{10 + j}, v{10 + j}, v{26 + j}, s[{40 + 2 * (j % 4)}:{41 + 2 * (j % 4)}]" for j in range(4, 8)])),
    ("min by v_min_f32: 4 v_pk_mov + 8 v_min_f32", 
     lambda k: "\n".join([f"\tv_pk_mov_b32 v[{26 + 2 * j}:{27 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}] op_sel:[0,1]" for j in range(4)] +
                         [f"\tv_min_f32 v{10 + j}, v{10 + j}, v{26 + j}" for j in range(8)])),
    ("v_minimum3_f32 independent", lambda k: f"\tv_minimum3_f32 v{10 + k % 16}, v40, v41, v41"),
    ("min by v_minimum3_f32: 4 v_pk_mov, zero test of a (3 v_min3 |a|, v_min, v_cmp, branch), 8 v_minimum3_f32",
     lambda k: "\n".join([f"\tv_pk_mov_b32 v[{26 + 2 * j}:{27 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}], v[{10 + 2 * j}:{11 + 2 * j}] op_sel:[0,1]" for j in range(4)] +
                         ["\tv_min3_f32 v34, |v10|, |v11|, |v12|", "\tv_min3_f32 v34, v34, |v13|, |v14|", "\tv_min3_f32 v34, v34, |v15|, |v16|", "\tv_min_f32 v34, v34, |v17|",
                          "\tv_cmp_eq_f32 vcc, 0, v34", "\ts_cbranch_vccnz .Lub_exit"] +
                         [f"\tv_minimum3_f32 v{10 + j}, v{10 + j}, v{26 + j}, v{26 + j}" for j in range(8)])),
]


def gen_ubench(a):
    name = "fh_ubench"
    a(f"""
	.text
	.protected {name}
	.globl {name}
	.p2align 8
	.type {name},@function
{name}:
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	s_load_dwordx2 s[6:7], s[0:1], 0x8
""" + "".join(f"\tv_mov_b32 v{r}, 1.0\n" for r in range(10, 40)) + f"""	v_mov_b32 v40, 1.0
	v_mov_b32 v41, 2.0
	v_mov_b32 v42, 1.0
	v_mov_b32 v43, 2.0
	v_mov_b32 v44, 1.0
	v_mov_b32 v45, 2.0
	s_mov_b32 s20, 0
	s_mov_b32 s21, 3
	s_mov_b32 s22, 0
	s_mov_b64 s[24:25], -1
	s_waitcnt lgkmcnt(0)
	s_mov_b32 s8, s7
	s_cmp_eq_u32 s8, 0
	s_cbranch_scc1 .Lub_exit""")
    for t in range(len(TESTS)):
        a(f"\ts_cmp_eq_u32 s6, {t}\n\ts_cbranch_scc1 .Lub_test{t}")
    a("\ts_branch .Lub_exit")
    for t, (desc, pat) in enumerate(TESTS):
        extra = ""
        if "ds_read" in desc:
            extra = "\tv_mov_b32 v10, 0\n\tv_mov_b32 v12, 0\n\tds_write_b32 v12, v12\n\ts_waitcnt lgkmcnt(0)"
        a(f"""
.Lub_test{t}:  ; {desc}
{extra}
	s_cmp_eq_u32 s20, 1          ; scc = 0
	s_memtime s[12:13]
	s_waitcnt lgkmcnt(0)
.Lub_loop{t}:""")
        if "setpc" in desc:   # s[30:31] = address of the first pattern
            a("\ts_getpc_b64 s[30:31]\n\ts_add_u32 s30, s30, 4")
        for k in range(32):
            a(pat(k))
        a(f"""
	s_cmp_eq_u32 s20, 0x7fffffff   ; keep scc = 0 for the not-taken test
	s_sub_u32 s8, s8, 1
	s_cmp_lg_u32 s8, 0
	s_cbranch_scc1 .Lub_loop{t}
	s_branch .Lub_done""")
    a(f"""
.Lub_done:
	s_set_gpr_idx_off
	s_memtime s[14:15]
	s_waitcnt lgkmcnt(0)
	s_sub_u32 s14, s14, s12
	s_subb_u32 s15, s15, s13
	v_cvt_f32_u32 v1, s14
	v_cvt_f32_u32 v2, s7
	v_mul_f32 v2, 0x42000000, v2
	v_rcp_f32 v2, v2
	s_nop 1
	v_mul_f32 v1, v1, v2
	v_lshlrev_b32 v3, 2, v0
	s_lshl_b32 s2, s2, 2
	v_mov_b32 v4, s2
	v_cmp_eq_u32 vcc, 0, v0
	s_and_saveexec_b64 s[16:17], vcc
	global_store_dword v4, v1, s[4:5]
.Lub_exit:
	s_endpgm
.L{name}_end:
	.size {name}, .L{name}_end - {name}
	.rodata
	.p2align 6
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 64
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 16
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_sgpr_workgroup_id_y 0
		.amdhsa_system_sgpr_workgroup_id_z 0
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 128
		.amdhsa_next_free_sgpr 64
		.amdhsa_accum_offset 128
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text""")
    return name, 16, 128, [(8, "global_buffer"), (4, "by_value"), (4, "by_value")]
