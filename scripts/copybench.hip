// copybench.hip -- measurement aid: streaming copy of a state array with the step kernels' exact layout and access pattern
// (component-major, 8 B/lane buffer loads/stores), in place and out of place, with the cache-policy hints and the
// XCD-contiguous workgroup order.  This is the memory system's ceiling the EKF step is compared with in DESIGN.md 6.
//   hipcc -O3 --offload-arch=gfx950 -o copybench scripts/copybench.hip && ./copybench > profiles/rNN_copybench.txt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mkbuf(const void*p, unsigned bytes){ return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p),0,bytes,0x00020000);}
template<int DEPTH, int LA, int SA, bool XCD> __global__ __launch_bounds__(64) void copyA(const double* src, double* dst, long stride, int B, int nc){
  unsigned wg=blockIdx.x;
  if(XCD){ unsigned nq=gridDim.x>>3,nr=gridDim.x&7u,x=blockIdx.x&7u,r=blockIdx.x>>3; wg=(x<nr? x*(nq+1u): nr*(nq+1u)+(x-nr)*nq)+r; }
  unsigned b=wg*64u+threadIdx.x; if(b>=(unsigned)B) return; unsigned bo=b*8u, s8=(unsigned)stride*8u;
  rsrc_t ri=mkbuf(src,(unsigned)nc*s8), ro=mkbuf(dst,(unsigned)nc*s8);
  for(int c0=0;c0<nc;c0+=DEPTH){ v2u v[DEPTH];
#pragma unroll
    for(int i=0;i<DEPTH;i++) v[i]=__builtin_amdgcn_raw_buffer_load_b64(ri,bo,(unsigned)(c0+i)*s8,LA);
#pragma unroll
    for(int i=0;i<DEPTH;i++){ v[i].x+=1u; __builtin_amdgcn_raw_buffer_store_b64(v[i],ro,bo,(unsigned)(c0+i)*s8,SA);}
  }
}

// 16 B per lane: component PAIRS interleaved, st[NC/2][stride][2] -- lane b reads components (2c, 2c+1) of its filter with one
// buffer_load_dwordx4 (1 KiB contiguous per wave instruction instead of 512 B)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template<int DEPTH, int LA, int SA, bool XCD, int WG> __global__ __launch_bounds__(WG) void copyB(const double* src, double* dst, long stride, int B, int npair){
  unsigned wg=blockIdx.x;
  if(XCD){ unsigned nq=gridDim.x>>3,nr=gridDim.x&7u,x=blockIdx.x&7u,r=blockIdx.x>>3; wg=(x<nr? x*(nq+1u): nr*(nq+1u)+(x-nr)*nq)+r; }
  unsigned b=wg*(unsigned)WG+threadIdx.x; if(b>=(unsigned)B) return; unsigned bo=b*16u, s16=(unsigned)stride*16u;
  rsrc_t ri=mkbuf(src,(unsigned)npair*s16), ro=mkbuf(dst,(unsigned)npair*s16);
  for(int c0=0;c0<npair;c0+=DEPTH){ v4u v[DEPTH];
#pragma unroll
    for(int i=0;i<DEPTH;i++) v[i]=__builtin_amdgcn_raw_buffer_load_b128(ri,bo,(unsigned)(c0+i)*s16,LA);
#pragma unroll
    for(int i=0;i<DEPTH;i++){ v[i].x+=1u; __builtin_amdgcn_raw_buffer_store_b128(v[i],ro,bo,(unsigned)(c0+i)*s16,SA);}
  }
}

// TILED layouts: st[B/TF][NC][TF] (8 B/lane) or st[B/TF][NC/2][TF][2] (16 B/lane): the whole state of TF filters is one
// contiguous block (TF=64: 71 680 B), so a wave's 140 component accesses stay inside one or two pages instead of touching
// 140 rows that lie `stride*8` bytes (8 MB at 1 M filters) apart
template<int DEPTH, int LA, int SA, bool XCD, int TF> __global__ __launch_bounds__(64) void copyT8(const double* src, double* dst, int B, int nc){
  unsigned wg=blockIdx.x;
  if(XCD){ unsigned nq=gridDim.x>>3,nr=gridDim.x&7u,x=blockIdx.x&7u,r=blockIdx.x>>3; wg=(x<nr? x*(nq+1u): nr*(nq+1u)+(x-nr)*nq)+r; }
  unsigned b=wg*64u+threadIdx.x; if(b>=(unsigned)B) return;
  const unsigned tile=b/TF, l=b%TF; const size_t tb=(size_t)tile*nc*TF*8;
  rsrc_t ri=mkbuf((const char*)src+tb,(unsigned)nc*TF*8), ro=mkbuf((char*)dst+tb,(unsigned)nc*TF*8);
  const unsigned bo=l*8u;
  for(int c0=0;c0<nc;c0+=DEPTH){ v2u v[DEPTH];
#pragma unroll
    for(int i=0;i<DEPTH;i++) v[i]=__builtin_amdgcn_raw_buffer_load_b64(ri,bo,(unsigned)(c0+i)*TF*8u,LA);
#pragma unroll
    for(int i=0;i<DEPTH;i++){ v[i].x+=1u; __builtin_amdgcn_raw_buffer_store_b64(v[i],ro,bo,(unsigned)(c0+i)*TF*8u,SA);}
  }
}
template<int DEPTH, int LA, int SA, bool XCD, int TF> __global__ __launch_bounds__(64) void copyT16(const double* src, double* dst, int B, int npair){
  unsigned wg=blockIdx.x;
  if(XCD){ unsigned nq=gridDim.x>>3,nr=gridDim.x&7u,x=blockIdx.x&7u,r=blockIdx.x>>3; wg=(x<nr? x*(nq+1u): nr*(nq+1u)+(x-nr)*nq)+r; }
  unsigned b=wg*64u+threadIdx.x; if(b>=(unsigned)B) return;
  const unsigned tile=b/TF, l=b%TF; const size_t tb=(size_t)tile*npair*TF*16;
  rsrc_t ri=mkbuf((const char*)src+tb,(unsigned)npair*TF*16), ro=mkbuf((char*)dst+tb,(unsigned)npair*TF*16);
  const unsigned bo=l*16u;
  for(int c0=0;c0<npair;c0+=DEPTH){ v4u v[DEPTH];
#pragma unroll
    for(int i=0;i<DEPTH;i++) v[i]=__builtin_amdgcn_raw_buffer_load_b128(ri,bo,(unsigned)(c0+i)*TF*16u,LA);
#pragma unroll
    for(int i=0;i<DEPTH;i++){ v[i].x+=1u; __builtin_amdgcn_raw_buffer_store_b128(v[i],ro,bo,(unsigned)(c0+i)*TF*16u,SA);}
  }
}
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
template<class F> float timeit(F f,int reps){ hipEvent_t a,b; hipEventCreate(&a);hipEventCreate(&b); f(); f(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<reps;i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms/reps; }
// 21-state tiles (129 rows of 64 x 16 B) with the step kernels' wave mappings: NW waves per tile, wave w moves rows
// [R[w], R[w+1]) -- every load of the wave first, then every store (what a role does around its arithmetic); rows beyond
// 64 per wave go in two halves (register file)
template<int NW, int LA, int SA, bool XCD> __global__ __launch_bounds__(NW*64) void copyQ(const double* src, double* dst, int B, int r0, int r1, int r2, int r3, int r4){
  unsigned wg=blockIdx.x;
  if(XCD){ unsigned nq=gridDim.x>>3,nr=gridDim.x&7u,x=blockIdx.x&7u,r=blockIdx.x>>3; wg=(x<nr? x*(nq+1u): nr*(nq+1u)+(x-nr)*nq)+r; }
  const unsigned lane=threadIdx.x&63u; const int w=__builtin_amdgcn_readfirstlane((int)(threadIdx.x>>6));
  const size_t tb=(size_t)wg*129*1024;
  rsrc_t ri=mkbuf((const char*)src+tb,129*1024), ro=mkbuf((char*)dst+tb,129*1024);
  const int lo_ = w==0? r0 : w==1? r1 : w==2? r2 : r3, hi_ = w==0? r1 : w==1? r2 : w==2? r3 : r4;
  const unsigned bo=lane*16u;
  for(int c0=lo_; c0<hi_; c0+=42){ v4u v[42];
#pragma unroll
    for(int i=0;i<42;i++) if(c0+i<hi_) v[i]=__builtin_amdgcn_raw_buffer_load_b128(ri,bo,(unsigned)(c0+i)*1024u,LA);
#pragma unroll
    for(int i=0;i<42;i++) if(c0+i<hi_){ v[i].x+=1u; __builtin_amdgcn_raw_buffer_store_b128(v[i],ro,bo,(unsigned)(c0+i)*1024u,SA); asm volatile("s_nop 1"::"v"(v[i])); }
  }
}

int main(){
  const int nc=140;
  for(int B : {32768, 65536, 196608, 262144, 1<<20}){
    long stride=B; size_t bytes=(size_t)nc*stride*8; double *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
    auto rep=[&](const char*name,float ms){ printf("B=%8d %-34s %9.1f us  %7.1f GB/s (r+w)\n",B,name,ms*1e3,2.0*bytes/(ms*1e-3)/1e9); fflush(stdout); };
    int reps = B>100000? 20: 100; int g=(B+63)/64;
#define RUN(NAME,D,LA,SA,X,DST) rep(NAME, timeit([&]{ copyA<D,LA,SA,X><<<g,64>>>(s,DST,stride,B,nc);},reps))
    RUN("out-of-place d35",35,0,0,false,d);
    RUN("out-of-place d35 xcd",35,0,0,true,d);
    RUN("in-place d35",35,0,0,false,s);
    RUN("in-place d35 xcd",35,0,0,true,s);
    RUN("in-place d35 xcd st-nt",35,0,2,true,s);
    RUN("in-place d35 xcd ld-nt st-nt",35,2,2,true,s);
    RUN("in-place d35 xcd ld-nt",35,2,0,true,s);
    RUN("in-place d35 xcd st-sc1",35,0,16,true,s);
    RUN("in-place d35 xcd st-sc0sc1",35,0,17,true,s);
    RUN("in-place d70 xcd",70,0,0,true,s);
    RUN("in-place d70 xcd st-nt",70,0,2,true,s);
#define RUNB(NAME,D,LA,SA,X,WG,DST) rep(NAME, timeit([&]{ copyB<D,LA,SA,X,WG><<<(B+WG-1)/WG,WG>>>(s,DST,stride,B,nc/2);},reps))
    RUNB("16B out-of-place d35",35,0,0,false,64,d);
    RUNB("16B in-place d35",35,0,0,false,64,s);
    RUNB("16B in-place d35 xcd",35,0,0,true,64,s);
    RUNB("16B in-place d35 xcd st-sc1",35,0,16,true,64,s);
    RUNB("16B in-place d35 xcd st-nt",35,0,2,true,64,s);
    RUNB("16B in-place d35 xcd ld-nt st-nt",35,2,2,true,64,s);
    RUNB("16B in-place d14 xcd",14,0,0,true,64,s);
    RUNB("16B in-place d14 xcd wg256",14,0,0,true,256,s);
    RUNB("16B in-place d14 xcd wg256 nt",14,2,2,true,256,s);
    RUNB("16B in-place d7 wg256",7,0,0,false,256,s);
    RUNB("16B in-place d7 wg256 ld-nt st-nt",7,2,2,false,256,s);
    RUNB("16B in-place d70 xcd",70,0,0,true,64,s);
    RUNB("16B in-place d70 xcd ld-nt st-nt",70,2,2,true,64,s);
#define RUNT8(NAME,D,LA,SA,X,TF,DST) rep(NAME, timeit([&]{ copyT8<D,LA,SA,X,TF><<<g,64>>>(s,DST,B,nc);},reps))
#define RUNT16(NAME,D,LA,SA,X,TF,DST) rep(NAME, timeit([&]{ copyT16<D,LA,SA,X,TF><<<g,64>>>(s,DST,B,nc/2);},reps))
    RUNT8("tile64 8B in-place d35",35,0,0,false,64,s);
    RUNT8("tile64 8B in-place d35 xcd",35,0,0,true,64,s);
    RUNT8("tile64 8B in-place d35 xcd st-sc1",35,0,16,true,64,s);
    RUNT8("tile64 8B in-place d35 xcd nt",35,2,2,true,64,s);
    RUNT8("tile64 8B in-place d35 nt",35,2,2,false,64,s);
    RUNT8("tile64 8B out-of-place d35 xcd",35,0,0,true,64,d);
    RUNT16("tile64 16B in-place d35",35,0,0,false,64,s);
    RUNT16("tile64 16B in-place d35 xcd",35,0,0,true,64,s);
    RUNT16("tile64 16B in-place d35 xcd st-sc1",35,0,16,true,64,s);
    RUNT16("tile64 16B in-place d35 xcd nt",35,2,2,true,64,s);
    RUNT16("tile64 16B in-place d35 nt",35,2,2,false,64,s);
    RUNT16("tile64 16B in-place d70 xcd",70,0,0,true,64,s);
    RUNT16("tile64 16B in-place d70 xcd nt",70,2,2,true,64,s);
    RUNT16("tile64 16B out-of-place d35 xcd",35,0,0,true,64,d);
    RUNT16("tile256 16B in-place d35 xcd",35,0,0,true,256,s);
    RUNT16("tile256 16B in-place d35 xcd nt",35,2,2,true,256,s);
    RUNT8("tile1024 8B in-place d35 xcd",35,0,0,true,1024,s);
    RUNT8("tile1024 8B in-place d35 xcd nt",35,2,2,true,1024,s);
    hipFree(s);hipFree(d);
  }
  for(int B : {32768, 65536, 131072, 262144}){
    size_t bytes=(size_t)(B/64)*129*1024; double *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
    auto rep=[&](const char*name,float ms){ printf("n21 B=%8d %-34s %9.1f us  %7.1f GB/s (r+w)\n",B,name,ms*1e3,2.0*bytes/(ms*1e-3)/1e9); fflush(stdout); };
    int reps = B>100000? 20: 100; int g=B/64;
#define RUNQ(NAME,NW,LA,SA,X,DST,R1,R2,R3) rep(NAME, timeit([&]{ copyQ<NW,LA,SA,X><<<g,NW*64>>>(s,DST,B,0,R1,R2,R3,129);},reps))
    RUNQ("4 waves (29|41|27|32 rows) xcd",4,0,0,true,s,29,70,97);
    RUNQ("4 waves xcd st-sc1",4,0,16,true,s,29,70,97);
    RUNQ("4 waves xcd nt",4,2,2,true,s,29,70,97);
    RUNQ("4 waves st-sc1",4,0,16,false,s,29,70,97);
    RUNQ("4 waves xcd st-sc1 out-of-place",4,0,16,true,d,29,70,97);
    RUNQ("2 waves (70|59 rows) xcd st-sc1",2,0,16,true,s,70,129,129);
    RUNQ("2 waves xcd nt",2,2,2,true,s,70,129,129);
    RUNQ("1 wave xcd st-sc1",1,0,16,true,s,129,129,129);
    hipFree(s);hipFree(d);
  }
  return 0;
}
