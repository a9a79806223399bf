"""Generates valu_rate.hip: one kernel per instruction form, each a loop of 16 asm blocks of 8 independent instructions
(8 accumulators per lane), run with 8 waves per SIMD; prints cycles of the nominal clock per wave64 instruction and SIMD.
    python gen_valu_rate.py > valu_rate.hip && hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate"""

# name -> (instruction template with {a} = the lane's accumulator, %8 / %9 = VGPR operands b / c, %10 = 64-bit SGPR mask, %11 = SGPR float; packed?)
OPS = {
    "fma": ("v_fma_f32 {a}, {a}, %8, %9", 1), "mul": ("v_mul_f32 {a}, {a}, %8", 1), "add": ("v_add_f32 {a}, {a}, %8", 1),
    "mov": ("v_mov_b32 {a}, %8", 1), "cndmask": ("v_cndmask_b32 {a}, {a}, %8, vcc", 1), "rcp": ("v_rcp_f32 {a}, {a}", 1),
    "sqrt": ("v_sqrt_f32 {a}, {a}", 1), "rsq": ("v_rsq_f32 {a}, {a}", 1), "sin": ("v_sin_f32 {a}, {a}", 1),
    "cmp": ("v_cmp_lt_f32 vcc, {a}, %8", 1), "max": ("v_max_f32 {a}, {a}, %8", 1), "fmaak": ("v_fmaak_f32 {a}, {a}, %8, 0x3f000000", 1),
    "mul_abs": ("v_mul_f32 {a}, |{a}|, %8", 1),
    "pk_fma": ("v_pk_fma_f32 {a}, {a}, %8, %9", 2), "pk_mul": ("v_pk_mul_f32 {a}, {a}, %8", 2), "pk_add": ("v_pk_add_f32 {a}, {a}, %8", 2),
    "cnd_sgpr": ("v_cndmask_b32_e64 {a}, {a}, %8, %10", 1), "fmac": ("v_fmac_f32 {a}, %8, %9", 1), "min": ("v_min_f32 {a}, {a}, %8", 1),
    "and": ("v_and_b32 {a}, {a}, %8", 1), "xor": ("v_xor_b32 {a}, {a}, %8", 1), "lshr": ("v_lshrrev_b32 {a}, 1, {a}", 1),
    "addu": ("v_add_u32 {a}, {a}, %8", 1), "cvt_i": ("v_cvt_i32_f32 {a}, {a}", 1), "rndne": ("v_rndne_f32 {a}, {a}", 1),
    "alignbit": ("v_alignbit_b32 {a}, {a}, %8, 7", 1), "bfi": ("v_bfi_b32 {a}, %8, {a}, %9", 1), "mul_neg": ("v_mul_f32 {a}, -{a}, %8", 1),
    "fma_abs": ("v_fma_f32 {a}, |{a}|, %8, -%9", 1), "med3": ("v_med3_f32 {a}, {a}, %8, %9", 1), "exp": ("v_exp_f32 {a}, {a}", 1),
    "log": ("v_log_f32 {a}, {a}", 1), "cos": ("v_cos_f32 {a}, {a}", 1), "fract": ("v_fract_f32 {a}, {a}", 1),
    # how many distinct VGPR sources an fp32 op reads
    "fma_same": ("v_fma_f32 {a}, {a}, {a}, {a}", 1), "fma_2src": ("v_fma_f32 {a}, {a}, %8, {a}", 1), "fma_sgpr": ("v_fma_f32 {a}, {a}, %11, %9", 1),
    "fma_sgpr2": ("v_fma_f32 {a}, {a}, %11, {a}", 1), "fma_const": ("v_fma_f32 {a}, {a}, 0.5, 1.0", 1), "fmac_sv": ("v_fmac_f32 {a}, %11, %9", 1),
    # round 5: candidates for the loop's selects, masks and combined tests (which of them issue at the full rate?)
    "bitop3": ("v_bitop3_b32 {a}, {a}, %8, %9 bitop3:0xca", 1), "bfe_i": ("v_bfe_i32 {a}, {a}, 0, 1", 1), "bfe_u": ("v_bfe_u32 {a}, {a}, 1, 1", 1),
    "ashr": ("v_ashrrev_i32 {a}, 31, {a}", 1), "lshl": ("v_lshlrev_b32 {a}, 30, {a}", 1), "max3": ("v_max3_f32 {a}, {a}, %8, %9", 1),
    "cmp_abs": ("v_cmp_lt_f32_e64 vcc, |{a}|, %8", 1), "cmp_class": ("v_cmp_class_f32 vcc, {a}, %8", 1), "and_or": ("v_and_or_b32 {a}, {a}, %8, %9", 1), "lshl_add": ("v_lshl_add_u32 {a}, {a}, 1, %8", 1), "lshl_or": ("v_lshl_or_b32 {a}, {a}, 1, %8", 1),
    "sub_co": ("v_subrev_co_u32 {a}, vcc, 1, {a}", 1), "cvt_f_u": ("v_cvt_f32_u32 {a}, {a}", 1), "cvt_ubyte": ("v_cvt_f32_ubyte0 {a}, {a}", 1),
    "ldexp": ("v_ldexp_f32 {a}, {a}, %8", 1), "cmp_u": ("v_cmp_eq_u32 vcc, {a}, %8", 1),
    "add3": ("v_add3_u32 {a}, {a}, %8, %9", 1), "perm": ("v_perm_b32 {a}, {a}, %8, %9", 1), "sub_f": ("v_sub_f32 {a}, {a}, %8", 1),
    "mul_sgpr": ("v_mul_f32 {a}, %11, {a}", 1), "mul_const": ("v_mul_f32 {a}, 0.5, {a}", 1), "mul_lit": ("v_mul_f32 {a}, 0x3f8ccccd, {a}", 1),
}
# instruction pairs / special sequences (8 lines each)
SEQUENCES = {
    "cnd_vcc_set": ["s_mov_b64 vcc, %10"] + [f"v_cndmask_b32 %{i}, %{i}, %8, vcc" for i in range(8)],
    "cnd_fma_mix": sum(([f"v_cndmask_b32_e64 %{i}, %{i}, %8, %10", f"v_fma_f32 %{i + 1}, %{i + 1}, %8, %9"] for i in range(0, 8, 2)), []),
    "cmp_cnd": sum(([f"v_cmp_lt_f32 vcc, %{i}, %9", f"v_cndmask_b32 %{i + 1}, %{i + 1}, %8, vcc"] for i in range(0, 8, 2)), []),
    "fma_chain4": [f"v_fma_f32 %{i % 4}, %{i % 4}, %8, %9" for i in range(8)],
    # does a source that is the previous instruction's result cost a register read?  ping-pong: each fma reads the one before
    "fma_fwd": [f"v_fma_f32 %{(i + 1) % 2}, %{i % 2}, %8, %9" for i in range(8)],
    "fma_fwd3": [f"v_fma_f32 %{(i + 1) % 2}, %{i % 2}, %{2 + i % 6}, %9" for i in range(8)],      # ... with a third rotating VGPR source
    "mul_fwd": [f"v_mul_f32 %{(i + 1) % 2}, %{i % 2}, %8" for i in range(8)],
    # same three distinct sources, register numbers chosen by the allocator (bank luck): 8 different destination/source sets
    "fma_3distinct": [f"v_fma_f32 %{i}, %{(i + 1) % 8}, %{(i + 2) % 8}, %{(i + 3) % 8}" for i in range(8)],
    "fma_2distinct_same": [f"v_fma_f32 %{i}, %{(i + 1) % 8}, %{(i + 1) % 8}, %{(i + 3) % 8}" for i in range(8)],
    "fmac_2src": [f"v_fmac_f32 %{i}, %{(i + 1) % 8}, %{(i + 1) % 8}" for i in range(8)],
}


def kernel(name, lines, width=1):
    block = "\\n".join(lines)
    t = "float" if width == 1 else "float2v"
    if width == 1:
        init = "".join(f"{t} a{i} = {{(float)threadIdx.x + {i}.f}};" for i in range(8))
        bc = f"{t} b = 1.0001f, c = 0.5f;"
        fin = "float r = " + " + ".join(f"a{i}" for i in range(8)) + ";"
    else:
        init = "".join(f"{t} a{i} = {{(float)threadIdx.x + {i}.f, {i}.5f}};" for i in range(8))
        bc = f"{t} b = {{1.0001f, 0.9999f}}, c = {{0.5f, 0.25f}};"
        fin = "float2v s = " + " + ".join(f"a{i}" for i in range(8)) + "; float r = s.x + s.y;"
    outs = ", ".join(f'"+v"(a{i})' for i in range(8))
    stmt = f'asm volatile("{block}" : {outs} : "v"(b), "v"(c), "s"(m), "s"(sb) : "vcc");'
    body = "\n".join("        " + stmt for _ in range(16))
    return f'''__global__ void __launch_bounds__(256) k_{name}(float* out, int iters, unsigned long long* clk) {{
    const unsigned long long born_cycles = __builtin_amdgcn_s_memtime(), born_ticks = __builtin_amdgcn_s_memrealtime();
    {init}
    {bc}
    unsigned long long m = 0x5555555555555555ull + (unsigned long long)iters;
    float sb = 1.0001f + (float)iters * 1e-9f;
    for (int i = 0; i < iters; i++) {{
{body}
    }}
    {fin}
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x % 64 == 0) {{   // every wave: its lifetime in shader cycles and in ticks of the constant 100 MHz clock
        atomicAdd(clk, (unsigned long long)__builtin_amdgcn_s_memtime() - born_cycles);
        atomicAdd(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - born_ticks);
    }}
}}'''


print("// generated by gen_valu_rate.py: VALU issue rates on gfx950 (cycles per wave64 instruction per SIMD at 8 waves/SIMD)")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\ntypedef float float2v __attribute__((ext_vector_type(2)));")
names = []
for name, (tmpl, width) in OPS.items():
    print(kernel(name, [tmpl.format(a=f"%{i}") for i in range(8)], width))
    names.append(name)
for name, lines in SEQUENCES.items():
    print(kernel(name, lines))
    names.append(name)
print('''template <typename K>
static void run(const char* name, K kernel, float* out, int blocks, unsigned long long* clk) {
    const int iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kernel<<<blocks, 256>>>(out, 64, clk);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 16);
    hipEventRecord(a);
    kernel<<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double wave_instr_per_simd = (double)blocks * 4 * iters * 128 / 1024.0;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double cycles = ms * 1e-3 * clk_khz * 1e3;
    unsigned long long c[2] = {0, 0};
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    double mhz = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;   // shader cycles per tick of the 100 MHz reference clock
    printf("%-12s %8.3f ms  %6.2f cycles of the nominal %d MHz clock per wave64 instruction per SIMD; shader clock while it ran %6.0f MHz -> %5.2f shader cycles\\n",
           name, ms, cycles / wave_instr_per_simd, clk_khz / 1000, mhz, ms * 1e-3 * mhz * 1e6 / wave_instr_per_simd);
}
int main() {
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned long long* clk;
    hipMalloc(&clk, 16);''')
for n in names:
    print(f'    run("{n}", k_{n}, out, blocks, clk);')
print("    return 0;\n}")
