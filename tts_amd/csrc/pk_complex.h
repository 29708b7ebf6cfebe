// Complex arithmetic on packed fp32 pairs for gfx950: a complex number lives in one aligned register pair and every
// helper is ONE v_pk_*_f32 instruction (two for a full complex product) whose operand swizzles (op_sel / op_sel_hi:
// which half of a source feeds the low / high lane) and per-lane negations do the work that would otherwise take
// register moves.  Written as inline assembly because the compiler's own vectoriser pairs up unrelated scalars and
// then shuffles them back (a fifth of the vector instructions of the Griffin-Lim kernel were moves).
#pragma once

typedef float f2 __attribute__((ext_vector_type(2)));

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 pk_addmj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 pk_addpj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// -a - i b = (-a.x + b.y, -a.y - b.x) and -a + i b = (-a.x - b.y, -a.y + b.x)
__device__ __forceinline__ f2 pk_neg_addmj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f2 pk_neg_addpj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// conj(a - i b) = (a.x + b.y, -a.y + b.x)
__device__ __forceinline__ f2 pk_conj_addmj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ f2 pk_addconj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f2 pk_subconj(f2 a, f2 b) {
  f2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// i a (CONJ: -i a)
template <bool CONJ>
__device__ __forceinline__ f2 pk_mulj(f2 a) {
  f2 d;
  if (!CONJ)
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,0]" : "=v"(d) : "v"(a));
  else
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a));
  return d;
}
// a * s.x and a * s.y (both lanes scaled by one half of s)
__device__ __forceinline__ f2 pk_scale_lo(f2 a, f2 s) {
  f2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(s));
  return d;
}
__device__ __forceinline__ f2 pk_scale_hi(f2 a, f2 s) {
  f2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(s));
  return d;
}
// a * w (CONJ: a * conj(w)): t = (a.y w.y, a.y w.x), then (a.x w.x -+ t.x, +-a.x w.y + t.y)
template <bool CONJ>
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 w) {
  f2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  if (!CONJ)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]"
        : "=v"(d) : "v"(a), "v"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,0] neg_hi:[1,0,0]"
        : "=v"(d) : "v"(a), "v"(w), "v"(t));
  return d;
}
// the same with a wave-uniform factor kept in scalar registers (constants)
template <bool CONJ>
__device__ __forceinline__ f2 pk_cmulc(f2 a, f2 w) {
  f2 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  if (!CONJ)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]"
        : "=v"(d) : "v"(a), "s"(w), "v"(t));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,0] neg_hi:[1,0,0]"
        : "=v"(d) : "v"(a), "s"(w), "v"(t));
  return d;
}
