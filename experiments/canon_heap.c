#include <math.h>
#include <stdint.h>
#include <stdlib.h>
typedef struct { float dist; uint32_t si; uint64_t val; } nd;
typedef struct { nd* a; size_t n, cap; } hp;
static int mode;
static inline int less(nd x, nd y){
  if (x.dist != y.dist) return x.dist < y.dist;
  if (mode==0){ if (x.val!=y.val) return x.val<y.val; return x.si<y.si; }
  if (mode==1){ if (x.si!=y.si) return x.si<y.si; return x.val<y.val; }
  if (mode==2){ if (x.si!=y.si) return x.si>y.si; return x.val<y.val; }
  return 0;
}
static void push(hp*h, nd x){ if(h->n==h->cap){h->cap=h->cap?h->cap*2:4096;h->a=realloc(h->a,h->cap*sizeof(nd));}
  size_t i=h->n++; while(i>0){size_t p=(i-1)/2; if(!less(x,h->a[p]))break; h->a[i]=h->a[p]; i=p;} h->a[i]=x;}
static nd pop(hp*h){ nd top=h->a[0]; nd x=h->a[--h->n]; size_t i=0; for(;;){size_t c=2*i+1; if(c>=h->n)break; if(c+1<h->n&&less(h->a[c+1],h->a[c]))c++; if(!less(h->a[c],x))break; h->a[i]=h->a[c]; i=c;} if(h->n)h->a[i]=x; return top;}
static const int8_t D[26][3]={{-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1},{-1,-1,0},{-1,1,0},{1,-1,0},{1,1,0},{0,-1,-1},{0,-1,1},{0,1,-1},{0,1,1},{-1,0,-1},{-1,0,1},{1,0,-1},{1,0,1},{-1,-1,-1},{1,-1,-1},{-1,1,-1},{-1,-1,1},{1,1,-1},{1,-1,1},{-1,1,1},{1,1,1}};
int64_t canon_ball(uint8_t* f,int64_t sx,int64_t sy,int64_t sz,float wx,float wy,float wz,const uint64_t* src,const float* md,int64_t ns,int m){
  mode=m; hp h={0,0,0}; int64_t sxy=sx*sy,cnt=0;
  for(int64_t i=0;i<ns;i++){nd x={0.0f,(uint32_t)i,src[i]};push(&h,x);}
  while(h.n){ nd t=pop(&h); if(!f[t.val])continue; f[t.val]=0;cnt++;
    uint64_t s=src[t.si]; float maxd=md[t.si];
    int64_t z=t.val/sxy,r=t.val%sxy,y=r/sx,x=r%sx; int64_t oz=s/sxy,orr=s%sxy,oy=orr/sx,ox=orr%sx;
    for(int i=0;i<26;i++){int64_t nx=x+D[i][0],ny=y+D[i][1],nz=z+D[i][2]; if(nx<0||ny<0||nz<0||nx>=sx||ny>=sy||nz>=sz)continue;
      int64_t q=nx+sx*ny+sxy*nz; if(!f[q])continue;
      float a=wx*(float)(nx-ox),b=wy*(float)(ny-oy),c=wz*(float)(nz-oz); float ss=a*a; float tt=b*b; float uu=c*c; ss=ss+tt; ss=ss+uu; float d=sqrtf(ss);
      if(d<maxd){nd n2={d,t.si,(uint64_t)q};push(&h,n2);} } }
  free(h.a); return cnt; }
