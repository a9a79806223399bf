# explicit-register ubench: how does the cost of an fp32 fma on gfx950 depend on WHICH registers its sources are?
# usage: python gen_operand_rate.py > operand_rate.hip && hipcc --offload-arch=gfx950 -O2 -w operand_rate.hip -o operand_rate
# 8 independent accumulators per lane (a = fma(a, b, c)), 8 waves per SIMD; b and c are two fixed registers or rotate
def pat(acc, b, c, ins="v_fma_f32"):
    return lambda i: (ins, acc[i], acc[i], b, c)
OLD = [8, 4, 7, 5, 6, 2, 1, 3]
CASES = {
    "old_exact_v9_v11": pat(OLD, 9, 11),
    "old_shift32": pat([r + 32 for r in OLD], 41, 43),
    "old_b10_c11": pat(OLD, 10, 11),
    "old_b9_c10": pat(OLD, 9, 10),
    "old_b12_c16": pat(OLD, 12, 16),
    "old_b13_c14": pat(OLD, 13, 14),
    "seq_acc_b9_c11": pat([1, 2, 3, 4, 5, 6, 7, 8], 9, 11),
    "acc16_23_b9_c11": pat(list(range(16, 24)), 9, 11),
    "acc16_23_b24_c25": pat(list(range(16, 24)), 24, 25),
    "acc16_23_b24_c26": pat(list(range(16, 24)), 24, 26),
    "acc16_23_b25_c27": pat(list(range(16, 24)), 25, 27),
    "fmac_old_exact": lambda i: ("v_fmac_f32", OLD[i], 9, 11, None),
    # b and c rotate over 4 registers each (no source register repeats in consecutive instructions)
    "rot_parity_a_nb_a": lambda i: ("v_fma_f32", 32 + i, 32 + i, 40 + (i + 1) % 4, 44 + (i + 2) % 4),
    "rot_parity_a_a_a": lambda i: ("v_fma_f32", 32 + i, 32 + i, 40 + i % 4, 44 + i % 4),
    "rot_parity_a_nb_nb": lambda i: ("v_fma_f32", 32 + i, 32 + i, 40 + (i + 1) % 4, 44 + (i + 1) % 4),
    "rot_dst_differs": lambda i: ("v_fma_f32", 48 + i, 32 + i, 40 + (i + 1) % 4, 44 + (i + 2) % 4),
}

def _rot(i): return f"v_fma_f32 v{48+i}, v{32+i}, v{40+(i+1)%4}, v{44+(i+2)%4}"
CASES.update({
    # what does one transcendental instruction cost inside a stream of plain fmas?  (8 instructions per block)
    "mix_8fma": [_rot(i) for i in range(8)],
    "mix_7fma_1rcp": [_rot(i) for i in range(7)] + ["v_rcp_f32 v55, v39"],
    "mix_6fma_2rcp": [_rot(i) for i in range(6)] + ["v_rcp_f32 v54, v38", "v_rcp_f32 v55, v39"],
    "mix_7fma_1sin": [_rot(i) for i in range(7)] + ["v_sin_f32 v55, v39"],
    "mix_7fma_1sqrt": [_rot(i) for i in range(7)] + ["v_sqrt_f32 v55, v39"],
    "mix_7fma_1rcp_used": [_rot(i) for i in range(6)] + ["v_rcp_f32 v55, v39", "v_fma_f32 v54, v55, v40, v45"],
    "mix_4fma_4rcp": [_rot(i) for i in range(4)] + [f"v_rcp_f32 v{52+i}, v{36+i}" for i in range(4)],
    "mix_7fma_1cmp": [_rot(i) for i in range(7)] + ["v_cmp_lt_f32 vcc, v39, v40"],
    "mix_7fma_1cndmask": [_rot(i) for i in range(7)] + ["v_cndmask_b32 v55, v39, v40, vcc"],
    "mix_7fma_1max": [_rot(i) for i in range(7)] + ["v_max_f32 v55, v39, v40"],
})
print("#include <hip/hip_runtime.h>\n#include <cstdio>")
for name, f in CASES.items():
    lines = []
    for i in range(8):
        if isinstance(f, list):
            lines.append(f[i]); continue
        ins, d, a, b, c = f(i)
        if ins == "v_fmac_f32":
            lines.append(f"{ins} v{d}, v{a}, v{b}")
        elif c is None:
            lines.append(f"{ins} v{d}, v{a}, v{b}")
        else:
            lines.append(f"{ins} v{d}, v{a}, v{b}, v{c}")
    block = "\\n".join(lines)
    clob = ", ".join([f'"v{r}"' for r in range(1, 56)] + ['"vcc"'])
    init = "\\n".join([f"v_mov_b32 v{r}, 0x3f800347" for r in range(1, 56)])
    fin = "\\n".join(["v_mov_b32 %0, v1"] + [f"v_add_f32 %0, %0, v{r}" for r in range(2, 56)])
    body = "\n".join(f'        asm volatile("{block}" ::: {clob});' for _ in range(16))
    print(f'''__global__ void __launch_bounds__(256) k_{name}(float* out, int iters) {{
    unsigned t = threadIdx.x;
    asm volatile("{init}" ::: {clob}); (void)t;
    for (int i = 0; i < iters; i++) {{
{body}
    }}
    float r;
    asm volatile("{fin}" : "=v"(r) :: {clob});
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}''')
print('''template <typename K> static void run(const char* name, K kernel, float* out, int blocks) {
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kernel<<<blocks, 256>>>(out, 64); hipDeviceSynchronize();
    hipEventRecord(a); kernel<<<blocks, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    int cus = 0, khz = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0); hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    double wave_instr_per_simd = double(blocks) * 4 * iters * 128.0 / (cus * 4.0);
    printf("%-34s %8.3f ms  %5.2f cycles per wave64 instruction per SIMD\\n", name, ms, ms * 1e-3 * khz * 1e3 / wave_instr_per_simd);
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; hipMalloc(&out, sizeof(float) * cus * 8 * 256);''')
for name in CASES:
    print(f'    run("{name}", k_{name}, out, cus * 8);')
print("    return 0;\n}")
