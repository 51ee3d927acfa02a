/*
 * kalign_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker; see kalign_oracle.h).
 *
 * Single-threaded C restatement of Kalign v3.5.1's progressive-alignment hot
 * path.  Citations are to /root/reference/lib/src.  The arithmetic is IEEE-754
 * binary32 in source order with no contraction (build with -ffp-contract=off):
 * the reference is compiled -O3 -mavx2 without FMA, so that evaluation order IS
 * the specification.
 *
 * Structure (deliberately not the reference's): the six forward/backward
 * functions of aln_seqseq.c / aln_seqprofile.c / aln_profileprofile.c are ONE
 * routine, ko_pass(), written over a direction-neutral column counter v
 * (v = 0 is the boundary column that carries the injected state, v = ncols the
 * far column), with the operand-specific score / gap terms behind small
 * accessors.  The HIP kernels use the same (u, v) formulation.
 *
 * Pinned against: oracle/_ref (the real library) via tests/golden/*.npz --
 * per-task coded paths, merged-profile hashes, top-level f/b row hashes,
 * meetup (meet, transition, score), final gap arrays.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "kalign_oracle.h"

#define F FLT_MAX
/* aln_seqseq.c:11-12 -- strict '>' so that ties keep the right-hand value */
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#define MAX3(a, b, c) MAX(MAX(a, b), c)

enum { K_SS = 0, K_SP = 1, K_PP = 2 };
enum { FWD = 0, BWD = 1 };

typedef struct { float a, ga, gb; } st;     /* aln_struct.h:9-14 */

typedef struct {
        int kind;
        const uint8_t* s1;      /* row residues (seq-seq) */
        const uint8_t* s2;      /* column residues (seq-seq, seq-profile) */
        const float* p1;        /* row profile, 64 floats x (len_a+2) */
        const float* p2;        /* column profile */
        int len_a, len_b;
        const float* subm;      /* 23x23 */
        float gpo, gpe, tgpe, soff;
        float sp_open, sp_ext, sp_text;   /* gpo*sip etc. (aln_seqprofile.c:31-33) */
        const float* bonus;
        int bstride;
        st* f;
        st* b;
        int* path;
        float msum;
        int mcount;
        /* refinement trials (aln_struct.h:32-35; round-robin mode of aln_seqseq.c:385-414) */
        float flip_threshold;
        int flip_trial, flip_stride, flip_counter;
        float* flip_margins;    /* adaptive budget: the baseline trial's margins (aln_refine.c:187-193) */
        int flip_margin_alloc;
} dp_t;

/* ---- gap / score terms --------------------------------- */

static inline float row_open(const dp_t* d, int rec) { return d->kind == K_SS ? -d->gpo  : d->p1[(rec << 6) + 27]; }
static inline float row_ext (const dp_t* d, int rec) { return d->kind == K_SS ? -d->gpe  : d->p1[(rec << 6) + 28]; }
static inline float row_text(const dp_t* d, int rec) { return d->kind == K_SS ? -d->tgpe : d->p1[(rec << 6) + 29]; }

static inline float col_open(const dp_t* d, int rec)
{
        if(d->kind == K_SS) return -d->gpo;
        if(d->kind == K_SP) return -d->sp_open;
        return d->p2[(rec << 6) + 27];
}
static inline float col_ext(const dp_t* d, int rec)
{
        if(d->kind == K_SS) return -d->gpe;
        if(d->kind == K_SP) return -d->sp_ext;
        return d->p2[(rec << 6) + 28];
}
static inline float col_text(const dp_t* d, int rec)
{
        if(d->kind == K_SS) return -d->tgpe;
        if(d->kind == K_SP) return -d->sp_text;
        return d->p2[(rec << 6) + 29];
}

/*
 * One linear-space Gotoh pass.
 *   FWD: aln_seqseq.c:15-119, aln_seqprofile.c:13-123, aln_profileprofile.c:17-156
 *   BWD: aln_seqseq.c:121-238, aln_seqprofile.c:125-230, aln_profileprofile.c:158-298
 * rows [r0, r1) (0-based positions of the row operand); columns window
 * [startb, endb] in s[]-index space.  Column counter v runs 0..ncols;
 * s-index j(v) = startb+v (FWD) or endb-v (BWD).  The column record scoring
 * s-index j is j (FWD, 1-based cell j scores position j-1) or j+1 (BWD).
 */
/* save / save_rows (prefix reuse, below): the row after `save_rows` rows goes to save[0 .. endb - startb] (column startb + k at k) */
static void ko_pass(dp_t* d, int dir, int r0, int r1, int startb, int endb, st* save, int save_rows)
{
        st* s = (dir == FWD) ? d->f : d->b;
        const int ncols = endb - startb;
        const int nrows = r1 - r0;
        const int near_terminal = (dir == FWD) ? (startb == 0) : (endb == d->len_b);
        const int far_terminal  = (dir == FWD) ? (endb == d->len_b) : (startb == 0);
        unsigned int nz[24];
        int v, u;

#define SJ(v_) ((dir == FWD) ? (startb + (v_)) : (endb - (v_)))
#define CREC(j_) ((dir == FWD) ? (j_) : ((j_) + 1))
#define CPREV(rec_) ((dir == FWD) ? ((rec_) - 1) : ((rec_) + 1))

        /* row "-1": the injected state in s[0], then a pure horizontal-gap run */
        s[SJ(0)] = s[0];
        for(v = 1; v < ncols; v++){
                const int j = SJ(v), jp = SJ(v - 1), rec = CREC(j);
                s[j].a = -F;
                if(!near_terminal){
                        s[j].ga = MAX(s[jp].ga + col_ext(d, rec), s[jp].a + col_open(d, rec));
                }else{
                        s[j].ga = MAX(s[jp].ga, s[jp].a) + col_text(d, rec);
                }
                s[j].gb = -F;
        }
        s[SJ(ncols)].a = -F; s[SJ(ncols)].ga = -F; s[SJ(ncols)].gb = -F;

        for(u = 0; u < nrows; u++){
                const int i = (dir == FWD) ? (r0 + u) : (r1 - 1 - u);
                const int rrec = i + 1;
                const int rprev = (dir == FWD) ? rrec - 1 : rrec + 1;
                const float o_row = row_open(d, rrec), e_row = row_ext(d, rrec), t_row = row_text(d, rrec);
                const float o_rowprev = row_open(d, rprev);
                const float* subp = NULL;
                const float* prow = NULL;
                int nnz = 0;
                float pa, pga, pgb, ca, xa, xga;
                const int j0 = SJ(0);

                if(d->kind == K_SS){
                        subp = d->subm + 23 * d->s1[i];
                }else{
                        prow = d->p1 + (rrec << 6);
                        if(d->kind == K_PP){
                                /* aln_profileprofile.c:70-77 */
                                for(int c = 0; c < 23; c++) if(prow[c]) nz[nnz++] = (unsigned int)c;
                        }
                }

                pa = s[j0].a; pga = s[j0].ga; pgb = s[j0].gb;
                s[j0].a = -F; s[j0].ga = -F;
                xa = -F; xga = -F;
                if(!near_terminal){
                        s[j0].gb = MAX(pgb + e_row, pa + o_row);
                }else{
                        s[j0].gb = MAX(pgb, pa) + t_row;
                }

                for(v = 1; v <= ncols; v++){
                        const int j = SJ(v), rec = CREC(j);
                        ca = s[j].a;
                        pa = MAX3(pa, pga + col_open(d, CPREV(rec)), pgb + o_rowprev);
                        if(d->kind == K_SS){
                                pa += subp[d->s2[rec - 1]] - d->soff;            /* aln_seqseq.c:82 */
                        }else if(d->kind == K_SP){
                                pa += prow[32 + d->s2[rec - 1]];                 /* aln_seqprofile.c:81 */
                        }else{
                                const float* pc = d->p2 + (rec << 6) + 32;
                                for(int c = nnz - 1; c >= 0; c--){               /* aln_profileprofile.c:101-107 */
                                        pa += prow[nz[c]] * pc[nz[c]];
                                }
                        }
                        if(d->bonus){
                                /* FWD indexes with the 1-based cell column, BWD with the
                                   0-based one (aln_seqseq.c:83-85 vs :199-201): both are j here */
                                pa += d->bonus[i * d->bstride + j];
                        }
                        s[j].a = pa;
                        if(v < ncols){
                                pga = s[j].ga;
                                s[j].ga = MAX(xga + col_ext(d, rec), xa + col_open(d, rec));
                                pgb = s[j].gb;
                                s[j].gb = MAX(pgb + e_row, ca + o_row);
                                pa = ca;
                                xa = s[j].a;
                                xga = s[j].ga;
                        }else{
                                s[j].ga = -F;
                                if(!far_terminal){
                                        s[j].gb = MAX(s[j].gb + e_row, ca + o_row);
                                }else{
                                        s[j].gb = MAX(s[j].gb, ca) + t_row;
                                }
                        }
                }
                if(save && u == save_rows - 1){
                        memcpy(save, s + startb, sizeof(st) * (size_t)(ncols + 1));
                }
        }
#undef SJ
#undef CREC
#undef CPREV
}

/*
 * Meetup at the Hirschberg split row (aln_seqseq.c:241-420, aln_seqprofile.c:232-415,
 * aln_profileprofile.c:301-483).  Candidate order per column: codes 1,2,3,5,6,7;
 * at the last column only 3 and 6.  First strictly-greater wins.
 */
typedef struct { float max, max2; int c, t, c2, t2; } best_t;

static inline void consider(best_t* b, float s, int i, int code)
{
        if(s > b->max){
                b->max2 = b->max; b->c2 = b->c; b->t2 = b->t;
                b->max = s; b->t = code; b->c = i;
        }else if(s > b->max2){
                b->max2 = s; b->c2 = i; b->t2 = code;
        }
}

static void ko_meetup(dp_t* d, int startb, int endb, int mid, int* meet, int* tr, float* score)
{
        const st* f = d->f;
        const st* b = d->b;
        best_t B = { -F, -F, -1, -1, -1, -1 };
        const float middle = (float)(endb - startb) / 2.0F + (float)startb;
        const int rrec = mid + 1;                 /* R  = P1[mid+1], R- = P1[mid] */
        const float g3 = row_open(d, rrec);
        const float g7 = row_open(d, rrec - 1);
        const float g6_near = (startb == 0) ? row_text(d, rrec) : row_ext(d, rrec);
        const float g6_far = (endb == d->len_b) ? row_text(d, rrec) : row_ext(d, rrec);
        float sub;
        int i;
        for(i = startb; i < endb; i++){
                sub = fabsf(middle - (float)i);
                sub /= 1000.0F;
                consider(&B, f[i].a + b[i].a - sub, i, 1);
                consider(&B, f[i].a + b[i].ga + col_open(d, i + 1) - sub, i, 2);
                consider(&B, f[i].a + b[i].gb + g3 - sub, i, 3);
                consider(&B, f[i].ga + b[i].a + col_open(d, i) - sub, i, 5);
                consider(&B, f[i].gb + b[i].gb + g6_near - sub, i, 6);
                consider(&B, f[i].gb + b[i].a + g7 - sub, i, 7);
        }
        i = endb;
        sub = fabsf(middle - (float)i);
        sub /= 1000.0F;
        consider(&B, f[i].a + b[i].gb + g3 - sub, i, 3);
        consider(&B, f[i].gb + b[i].gb + g6_far - sub, i, 6);

        if(B.max2 > -F){                          /* aln_seqseq.c:376-383 */
                if(d->flip_margins && d->mcount < d->flip_margin_alloc) d->flip_margins[d->mcount] = B.max - B.max2;
                d->msum += B.max - B.max2;
                d->mcount++;
        }
        /* perturbation of refinement trials, round-robin mode (aln_seqseq.c:385-414): an uncertain meetup -- margin
           below the threshold -- takes the second-best choice when its running number falls on this trial's slot */
        if(d->flip_threshold > 0.0F && B.c2 >= 0 && B.max2 > -F){
                const float margin = B.max - B.max2;
                if(margin < d->flip_threshold){
                        if(d->flip_trial > 0 && d->flip_counter % d->flip_stride == d->flip_trial - 1){
                                B.c = B.c2; B.t = B.t2;
                        }
                        d->flip_counter++;
                }
        }
        *meet = B.c; *tr = B.t; *score = B.max;
}

/*
 * Hirschberg driver: aln_runner_serial + aln_continue (aln_controller.c:122-436).
 * fin / bin are the boundary states injected into f[0] / b[0].
 */
typedef struct { int top_meet, top_tr; float top_score; int have_top; uint64_t fhash, bhash; float* f_out; float* b_out; } probe_t;

static const st Z  = { 0.0F, -F, -F };
static const st GA = { -F, 0.0F, -F };
static const st GB = { -F, -F, 0.0F };

/*
 * Hirschberg PREFIX REUSE (the device's rule, kalign_amd/csrc/ka_meetup.h; switched on with ko_set_prefix_reuse -- the
 * reference always runs both passes): a child shares one corner with its parent.  The top-left child ([starta, mid') x
 * [startb, meet']) starts its forward pass from the parent's forward start state, so its forward rows are the parent's
 * forward rows restricted to its columns -- except in its LAST column, where a pass writes ga = -FLT_MAX
 * (aln_seqseq.c:108-117, aln_profileprofile.c:128-151) while the parent computed an inner ga there; the terminal rule of that
 * column (endb == len_b) can only differ when the child's last column is not the parent's, and then it is off on both
 * sides.  The bottom-right child shares the backward corner the same way.  A pass that RUNS therefore leaves the row the
 * usual child will want -- after (n - 1) / 2 of its n rows going forward (child rows [starta, mid - 1)), after n / 2
 * going backward (child rows [mid + 1, enda)) -- and a child whose own pass would have exactly that many rows takes the
 * row (last column's ga overwritten) instead of running it.  A pass that was itself taken over has left nothing
 * (one level deep: what the kernels do).  Executed cells: ~1.58 x rows x columns instead of 2 x.
 */
static int ko_reuse_on = 0;
static long long ko_cells_run = 0, ko_cells_reused = 0;
void ko_set_prefix_reuse(int on) { ko_reuse_on = on; ko_cells_run = 0; ko_cells_reused = 0; }
void ko_prefix_reuse_cells(long long* run, long long* reused) { *run = ko_cells_run; *reused = ko_cells_reused; }

typedef struct { const st* row; int rows; int startb; } ko_saved;     /* a parent's saved row: after `rows` rows, column startb at [0] */

static void ko_hirschberg_r(dp_t* d, int starta, int enda, int startb, int endb, st fin, st bin, probe_t* probe, ko_saved pf, ko_saved pb);

static void ko_hirschberg(dp_t* d, int starta, int enda, int startb, int endb, st fin, st bin, probe_t* probe)
{
        const ko_saved none = { NULL, 0, 0 };
        ko_hirschberg_r(d, starta, enda, startb, endb, fin, bin, probe, none, none);
}

static void ko_hirschberg_r(dp_t* d, int starta, int enda, int startb, int endb, st fin, st bin, probe_t* probe, ko_saved pf, ko_saved pb)
{
        const ko_saved none = { NULL, 0, 0 };
        ko_saved sf = none, sb = none;                  /* what THIS node's passes leave for its children */
        st* keep_f = NULL;
        st* keep_b = NULL;
        int mid, meet, tr;
        float score;
        if(starta >= enda) return;
        if(startb >= endb) return;
        mid = ((enda - starta) / 2) + starta;
        d->f[0] = fin;
        d->b[0] = bin;
        {
                const int nf = mid - starta, nb = enda - mid, ncols = endb - startb;
                const int rs_f = (nf - 1) / 2, rs_b = nb / 2;
                if(ko_reuse_on && pf.row && nf >= 1 && pf.rows == nf){
                        memcpy(d->f + startb, pf.row + (startb - pf.startb), sizeof(st) * (size_t)(ncols + 1));
                        d->f[endb].ga = -F;
                        ko_cells_reused += (long long)nf * ncols;
                }else{
                        if(ko_reuse_on && rs_f >= 1){
                                keep_f = (st*)malloc(sizeof(st) * (size_t)(ncols + 1));
                                sf.row = keep_f; sf.rows = rs_f; sf.startb = startb;
                        }
                        ko_pass(d, FWD, starta, mid, startb, endb, keep_f, rs_f);
                        ko_cells_run += (long long)nf * ncols;
                }
                if(ko_reuse_on && pb.row && nb >= 1 && pb.rows == nb){
                        memcpy(d->b + startb, pb.row + (startb - pb.startb), sizeof(st) * (size_t)(ncols + 1));
                        d->b[startb].ga = -F;
                        ko_cells_reused += (long long)nb * ncols;
                }else{
                        if(ko_reuse_on && rs_b >= 1){
                                keep_b = (st*)malloc(sizeof(st) * (size_t)(ncols + 1));
                                sb.row = keep_b; sb.rows = rs_b; sb.startb = startb;
                        }
                        ko_pass(d, BWD, mid, enda, startb, endb, keep_b, rs_b);
                        ko_cells_run += (long long)nb * ncols;
                }
        }
        ko_meetup(d, startb, endb, mid, &meet, &tr, &score);
        if(probe && !probe->have_top){
                probe->have_top = 1;
                probe->top_meet = meet; probe->top_tr = tr; probe->top_score = score;
                probe->fhash = ko_fnv1a(d->f, sizeof(st) * (uint64_t)(endb + 1));
                probe->bhash = ko_fnv1a(d->b, sizeof(st) * (uint64_t)(endb + 1));
                if(probe->f_out) memcpy(probe->f_out, d->f, sizeof(st) * (size_t)(endb + 1));
                if(probe->b_out) memcpy(probe->b_out, d->b, sizeof(st) * (size_t)(endb + 1));
        }
        switch(tr){
        case 1:
                d->path[mid] = meet; d->path[mid + 1] = meet + 1;
                ko_hirschberg_r(d, starta, mid - 1, startb, meet - 1, fin, Z, probe, sf, none);
                ko_hirschberg_r(d, mid + 1, enda, meet + 1, endb, Z, bin, probe, none, sb);
                break;
        case 2:
                d->path[mid] = meet;
                ko_hirschberg_r(d, starta, mid - 1, startb, meet - 1, fin, Z, probe, sf, none);
                ko_hirschberg_r(d, mid, enda, meet + 1, endb, GA, bin, probe, none, sb);
                break;
        case 3:
                d->path[mid] = meet;
                ko_hirschberg_r(d, starta, mid - 1, startb, meet - 1, fin, Z, probe, sf, none);
                ko_hirschberg_r(d, mid + 1, enda, meet, endb, GB, bin, probe, none, sb);
                break;
        case 5:
                d->path[mid + 1] = meet + 1;
                ko_hirschberg_r(d, starta, mid, startb, meet - 1, fin, GA, probe, sf, none);
                ko_hirschberg_r(d, mid + 1, enda, meet + 1, endb, Z, bin, probe, none, sb);
                break;
        case 6:
                ko_hirschberg_r(d, starta, mid - 1, startb, meet, fin, GB, probe, sf, none);
                ko_hirschberg_r(d, mid + 1, enda, meet, endb, GB, bin, probe, none, sb);
                break;
        case 7:
                d->path[mid + 1] = meet + 1;
                ko_hirschberg_r(d, starta, mid - 1, startb, meet, fin, GB, probe, sf, none);
                ko_hirschberg_r(d, mid + 1, enda, meet + 1, endb, Z, bin, probe, none, sb);
                break;
        default:
                break;
        }
        if(keep_f) free(keep_f);
        if(keep_b) free(keep_b);
}

/* init_alnmem (aln_setup.c:13-38) + run; d->f, d->b, d->path must hold
   max(len_a,len_b)+2 states / len_a+len_b+2 ints. */
static void ko_align(dp_t* d, probe_t* probe)
{
        const int g = (d->len_a > d->len_b ? d->len_a : d->len_b) + 2;
        for(int i = 0; i < g; i++) d->path[i] = -1;
        d->msum = 0.0F; d->mcount = 0;
        ko_hirschberg(d, 0, d->len_a, 0, d->len_b, Z, Z, probe);
}

uint64_t ko_fnv1a(const void* p, uint64_t n)
{
        const unsigned char* c = (const unsigned char*)p;
        uint64_t h = 1469598103934665603ULL;
        for(uint64_t i = 0; i < n; i++){ h ^= c[i]; h *= 1099511628211ULL; }
        return h;
}

/* ---- per-task glue --------------------------------------------------------------- */

/* make_profile_n (aln_setup.c:40-99), weight 1.0; prof holds (len+2)*64 floats */
int ko_make_profile(const uint8_t* seq, int len, const float* subm,
                    float gpo, float gpe, float tgpe, float soff, float* prof)
{
        memset(prof, 0, sizeof(float) * 64 * (size_t)(len + 2));
        for(int r = 0; r <= len + 1; r++){
                float* p = prof + ((size_t)r << 6);
                if(r >= 1 && r <= len){
                        const int c = seq[r - 1];
                        p[c] += 1.0F;
                        for(int j = 0; j < 23; j++) p[32 + j] = subm[23 * c + j] - soff;
                }
                p[55] = -gpo; p[56] = -gpe; p[57] = -tgpe;
        }
        return 0;
}

/* set_gap_penalties_n (aln_setup.c:101-119) */
int ko_set_gap_penalties(float* prof, int len, int nsip)
{
        for(int r = 0; r <= len + 1; r++){
                float* p = prof + ((size_t)r << 6);
                p[27] = p[55] * (float)nsip;
                p[28] = p[56] * (float)nsip;
                p[29] = p[57] * (float)nsip;
        }
        return 0;
}

/* mirror_path_n (aln_setup.c:438-462): raw_in is indexed by DP rows (= b
   positions, 1..len_b) and holds a-columns; raw_out is indexed by a positions. */
int ko_mirror_path(const int* raw_in, int len_a, int len_b, int* raw_out)
{
        for(int i = 0; i < len_a + 2; i++) raw_out[i] = -1;
        for(int i = 1; i <= len_b; i++) if(raw_in[i] != -1) raw_out[raw_in[i]] = i;
        return 0;
}

/*
 * add_gap_info_to_path_n (aln_setup.c:121-228).  Ops: 0 match, 1 gap in a,
 * 2 gap in b; coded[0] = alignment length, coded[alnlen+1] = 3.
 * NOTE (as executed, not as intended): the open/extend/close flag loop at
 * aln_setup.c:193-207 tests `o_path[j] != 3` with j parked on the terminator it
 * has just written, so it never runs; bits 4/8/16 are never set.  Only the
 * terminal-run flag 32 (:209-219) is applied.  We restate what executes.
 * coded must hold len_a+len_b+2 ints.
 */
int ko_code_path(const int* raw, int len_a, int len_b, int* coded)
{
        int j = 1, prev;
        for(int i = 0; i < len_a + len_b + 2; i++) coded[i] = 0;
        if(raw[1] == -1){
                coded[j++] = 2;
        }else{
                for(int k = 0; k < raw[1] - 1; k++) coded[j++] = 1;
                coded[j++] = 0;
        }
        prev = raw[1];
        for(int i = 2; i <= len_a; i++){
                if(raw[i] == -1){
                        coded[j++] = 2;
                }else{
                        if(raw[i] - 1 != prev && prev != -1){
                                for(int k = 0; k < raw[i] - prev - 1; k++) coded[j++] = 1;
                        }
                        coded[j++] = 0;
                }
                prev = raw[i];
        }
        if(raw[len_a] < len_b && raw[len_a] != -1){
                for(int k = 0; k < len_b - raw[len_a]; k++) coded[j++] = 1;
        }
        coded[0] = j - 1;
        coded[j] = 3;
        {
                int i = 1;
                while(coded[i] != 0){ coded[i] |= 32; i++; }
                i = coded[0];
                while(coded[i] != 0){ coded[i] |= 32; i--; }
        }
        return 0;
}

/*
 * convert_raw_path (aln_refine.c:591-672): the path coding of the refinement pass.  Unlike add_gap_info_to_path_n it
 * fills the gap-in-a run before a match from the last MATCHED column, and its flag loop does execute: 4 = first op of
 * a gap run, 8 = continuation, 16 (or +8 when it already carries 8) = last op before a match, 32 = terminal runs.
 */
int ko_convert_raw_path(const int* raw, int len_a, int len_b, int* o)
{
        int j = 1, b_last = 0, i;
        for(i = 0; i < len_a + len_b + 2; i++) o[i] = 0;
        for(i = 1; i <= len_a; i++){
                if(raw[i] == -1){
                        o[j++] = 2;
                }else{
                        for(int a = b_last + 1; a < raw[i]; a++) o[j++] = 1;
                        o[j++] = 0;
                        b_last = raw[i];
                }
        }
        for(int a = b_last + 1; a <= len_b; a++) o[j++] = 1;
        o[0] = j - 1;
        o[j] = 3;
        i = 2;
        while(o[i] != 3){
                if((o[i - 1] & 3) && !(o[i] & 3)){
                        if(o[i - 1] & 8) o[i - 1] += 8; else o[i - 1] |= 16;
                }else if(!(o[i - 1] & 3) && (o[i] & 3)){
                        o[i] |= 4;
                }else if((o[i - 1] & 1) && (o[i] & 1)){
                        o[i] |= 8;
                }else if((o[i - 1] & 2) && (o[i] & 2)){
                        o[i] |= 8;
                }
                i++;
        }
        i = 1;
        while(o[i] != 0){ o[i] |= 32; i++; }
        i = o[0];
        while(o[i] != 0){ o[i] |= 32; i--; }
        return 0;
}

/*
 * compute_sp_score (sp_score.c:75-201): profile-based sum-of-pairs score of the cross-group pairs along a coded
 * path -- integer residue / gap counts per column of both groups (build_profile, :22-58, from the members' residues
 * and their gaps[] BEFORE this merge), then one sequential fp32 walk: substitution terms in (i, j) order, gap terms.
 */
static void sp_build_profile(const uint8_t* codes, const int* off, const int* lens, int** gaps,
                             const int* sip, int nsip, int prof_len, int* freq, int* n_gap)
{
        for(int m = 0; m < nsip; m++){
                const int si = sip[m];
                int pos = 0;
                for(int j = 0; j < lens[si]; j++){
                        for(int k = 0; k < gaps[si][j]; k++){ n_gap[pos]++; pos++; }
                        const int r = codes[off[si] + j];
                        if(r < 23) freq[pos * 23 + r]++; else n_gap[pos]++;
                        pos++;
                }
                for(int k = 0; k < gaps[si][lens[si]]; k++){ n_gap[pos]++; pos++; }
                (void)prof_len;
        }
}

static float ko_sp_score(const uint8_t* codes, const int* off, const int* lens, int** gaps, const int* path,
                         const int* sip_a, int nsip_a, const int* sip_b, int nsip_b,
                         const float* subm, float gpo, float gpe, float tgpe)
{
        int pla = lens[sip_a[0]], plb = lens[sip_b[0]];
        for(int i = 0; i <= lens[sip_a[0]]; i++) pla += gaps[sip_a[0]][i];
        for(int i = 0; i <= lens[sip_b[0]]; i++) plb += gaps[sip_b[0]][i];
        int* fa = calloc((size_t)pla * 23 + 1, sizeof(int));
        int* ga = calloc((size_t)pla + 1, sizeof(int));
        int* fb = calloc((size_t)plb * 23 + 1, sizeof(int));
        int* gb = calloc((size_t)plb + 1, sizeof(int));
        float total = 0.0F;
        int pos_a = 0, pos_b = 0, in_a = 0, in_b = 0;
        sp_build_profile(codes, off, lens, gaps, sip_a, nsip_a, pla, fa, ga);
        sp_build_profile(codes, off, lens, gaps, sip_b, nsip_b, plb, fb, gb);
        for(int c = 1; c <= path[0]; c++){
                const int step = path[c] & 3;
                const float pen = (path[c] & 32) ? tgpe : gpe;
                if(step == 0){
                        const int* xa = fa + pos_a * 23;
                        const int* xb = fb + pos_b * 23;
                        for(int i = 0; i < 23; i++){
                                if(xa[i] == 0) continue;
                                for(int j = 0; j < 23; j++){
                                        if(xb[j] == 0) continue;
                                        total += (float)(xa[i] * xb[j]) * subm[i * 23 + j];
                                }
                        }
                        {
                                const int n_res_a = nsip_a - ga[pos_a], n_gap_b = gb[pos_b];
                                const int n_gap_a = ga[pos_a], n_res_b = nsip_b - gb[pos_b];
                                total -= (float)(n_res_a * n_gap_b + n_gap_a * n_res_b) * pen;
                        }
                        in_a = 0; in_b = 0; pos_a++; pos_b++;
                }else if(step == 1){
                        const int n_pairs = nsip_a * (nsip_b - gb[pos_b]);
                        if(!in_a) total -= (float)n_pairs * gpo;
                        total -= (float)n_pairs * pen;
                        in_a = 1; in_b = 0; pos_b++;
                }else if(step == 2){
                        const int n_pairs = (nsip_a - ga[pos_a]) * nsip_b;
                        if(!in_b) total -= (float)n_pairs * gpo;
                        total -= (float)n_pairs * pen;
                        in_a = 0; in_b = 1; pos_a++;
                }
        }
        free(fa); free(ga); free(fb); free(gb);
        return total;
}

/*
 * update_n (aln_setup.c:230-436): merged profile along a coded path.
 * out holds (coded[0]+2)*64 floats.
 */
static void gap_column(float* np, const float* src, int code, float sip_absent,
                       float gpo, float gpe, float tgpe)
{
        float gp;
        for(int i = 64; i--;) np[i] = src[i];
        if(!(code & 20)){
                if(code & 32){ np[25] += sip_absent; gp = tgpe * sip_absent; }
                else         { np[24] += sip_absent; gp = gpe * sip_absent; }
                for(int j = 32; j < 55; j++) np[j] -= gp;
        }else{
                if(code & 16){
                        if(code & 32){
                                np[25] += sip_absent; gp = tgpe * sip_absent;
                                np[23] += sip_absent; gp += gpo * sip_absent;
                        }else{
                                np[23] += sip_absent; gp = gpo * sip_absent;
                        }
                        for(int j = 32; j < 55; j++) np[j] -= gp;
                }
                if(code & 4){
                        if(code & 32){
                                np[25] += sip_absent; gp = tgpe * sip_absent;
                                np[23] += sip_absent; gp += gpo * sip_absent;
                        }else{
                                np[23] += sip_absent; gp = gpo * sip_absent;
                        }
                        for(int j = 32; j < 55; j++) np[j] -= gp;
                }
        }
}

static void sum_column(float* np, const float* pa, const float* pb, int rebalance, float sA, float sB)
{
        if(rebalance){
                for(int i = 0; i < 23; i++) np[i] = pa[i] * sA + pb[i] * sB;
                for(int i = 23; i < 64; i++) np[i] = pa[i] + pb[i];
        }else{
                for(int i = 64; i--;) np[i] = pa[i] + pb[i];
        }
}

int ko_update_profile(const float* pa, const float* pb, float* out, const int* coded,
                      int sipa, int sipb, const float* subm,
                      float gpo, float gpe, float tgpe, float use_seq_weights)
{
        float sA = 1.0f, sB = 1.0f;
        int rebalance = 0;
        int c = 1;
        if(use_seq_weights > 0.0f && sipa > 0 && sipb > 0){          /* aln_setup.c:255-262 */
                const float pseudo = use_seq_weights;
                const float total = (float)(sipa + sipb);
                const float denom = total + 2.0f * pseudo;
                sA = total * ((float)sipa + pseudo) / (denom * (float)sipa);
                sB = total * ((float)sipb + pseudo) / (denom * (float)sipb);
                rebalance = 1;
        }
        sum_column(out, pa, pb, rebalance, sA, sB);
        pa += 64; pb += 64; out += 64;
        while(coded[c] != 3){
                const int code = coded[c];
                if(!code){
                        sum_column(out, pa, pb, rebalance, sA, sB);
                        if(rebalance){                                   /* aln_setup.c:289-302 */
                                const float dA = sA - 1.0f, dB = sB - 1.0f;
                                for(int j = 0; j < 23; j++){
                                        float delta = 0.0f;
                                        for(int aa = 0; aa < 23; aa++){
                                                delta += (pa[aa] * dA + pb[aa] * dB) * subm[23 * aa + j];
                                        }
                                        out[32 + j] += delta;
                                }
                        }
                        pa += 64; pb += 64;
                }
                if(code & 1){
                        gap_column(out, pb, code, (float)sipa, gpo, gpe, tgpe);
                        pb += 64;
                }
                if(code & 2){
                        gap_column(out, pa, code, (float)sipb, gpo, gpe, tgpe);
                        pa += 64;
                }
                out += 64;
                c++;
        }
        sum_column(out, pa, pb, rebalance, sA, sB);
        return 0;
}

/* make_seq + update_gaps (weave_alignment.c:41-112) */
static void fold_gaps(int len, int* gis, const int* newgaps)
{
        int rel = 0;
        for(int i = 0; i <= len; i++){
                int add = 0;
                for(int j = rel; j <= rel + gis[i]; j++) add += newgaps[j];
                rel += gis[i] + 1;
                gis[i] += add;
        }
}

static void weave(const int* coded, const int* mem_a, int na, const int* mem_b, int nb,
                  const int* lens, int** gaps)
{
        const int n = coded[0] + 1;
        int* ga = calloc((size_t)n, sizeof(int));
        int* gb = calloc((size_t)n, sizeof(int));
        int posa = 0, posb = 0;
        for(int c = 1; coded[c] != 3; c++){
                if(!coded[c]){ posa++; posb++; }
                else if(coded[c] & 1){ ga[posa] += 1; posb++; }
                else if(coded[c] & 2){ gb[posb] += 1; posa++; }
        }
        for(int i = na; i--;) fold_gaps(lens[mem_a[i]], gaps[mem_a[i]], ga);
        for(int i = nb; i--;) fold_gaps(lens[mem_b[i]], gaps[mem_b[i]], gb);
        free(ga); free(gb);
}

/* mean of seq_distances over both clusters, in sip order (aln_run.c:126-203) */
static float mean_distance(const float* dist, const int* ma, int na, const int* mb, int nb, int numseq, int* count)
{
        float sum = 0.0f;
        int n = 0;
        for(int i = 0; i < na; i++) if(ma[i] < numseq){ sum += dist[ma[i]]; n++; }
        for(int i = 0; i < nb; i++) if(mb[i] < numseq){ sum += dist[mb[i]]; n++; }
        *count = n;
        return n ? sum / (float)n : 0.0f;
}


/* ---- anchor consistency (anchor_consistency.c) ------------------------------------- */
typedef struct { int K, N; float weight; int* anchor_ids; int** maps; } ko_cons;

/* farthest-first on |dist[i] - dist[anchor]| (anchor_consistency.c:122-192) */
static void ko_select_anchors(const float* dist, int N, int K, int* ids)
{
        float* min_dist = malloc(sizeof(float) * (size_t)N);
        float sum = 0.0f, mean, best_diff = FLT_MAX;
        int best = 0;
        for(int i = 0; i < N; i++) sum += dist[i];
        mean = sum / (float)N;
        for(int i = 0; i < N; i++){
                float diff = dist[i] - mean;
                if(diff < 0) diff = -diff;
                if(diff < best_diff){ best_diff = diff; best = i; }
        }
        ids[0] = best;
        for(int i = 0; i < N; i++){
                float d = dist[i] - dist[ids[0]];
                if(d < 0) d = -d;
                min_dist[i] = d;
        }
        for(int k = 1; k < K; k++){
                float best_min = -1.0f;
                best = 0;
                for(int i = 0; i < N; i++){
                        int skip = 0;
                        for(int j = 0; j < k; j++) if(ids[j] == i){ skip = 1; break; }
                        if(skip) continue;
                        if(min_dist[i] > best_min){ best_min = min_dist[i]; best = i; }
                }
                ids[k] = best;
                for(int i = 0; i < N; i++){
                        float d = dist[i] - dist[best];
                        if(d < 0) d = -d;
                        if(d < min_dist[i]) min_dist[i] = d;
                }
        }
        free(min_dist);
}

/* coded path -> position map (anchor_consistency.c:86-114) */
static void ko_posmap(const int* coded, int len_i, int* map)
{
        int pos_a = 0, pos_b = 0;
        for(int c = 0; c < len_i; c++) map[c] = -1;
        for(int c = 1; coded[c] != 3; c++){
                if(coded[c] == 0){
                        if(pos_a < len_i) map[pos_a] = pos_b;
                        pos_a++; pos_b++;
                }else if(coded[c] & 1){
                        pos_b++;
                }else if(coded[c] & 2){
                        if(pos_a < len_i) map[pos_a] = -1;
                        pos_a++;
                }
        }
}

static void ko_cons_free(ko_cons* ct)
{
        if(!ct) return;
        if(ct->maps){ for(int i = 0; i < ct->N * ct->K; i++) free(ct->maps[i]); free(ct->maps); }
        free(ct->anchor_ids);
        free(ct);
}

/* anchor_consistency_build (anchor_consistency.c:194-275): NULL when it declines (K<=0, N<3, no distances) */
static ko_cons* ko_cons_build(int N, const uint8_t* codes, const int* off, const int* lens, const float* dist,
                              const float* subm, float gpo, float gpe, float tgpe, int K, float weight)
{
        ko_cons* ct;
        if(K <= 0 || N < 3 || !dist) return NULL;
        if(K > N) K = N;
        ct = calloc(1, sizeof(ko_cons));
        ct->K = K; ct->N = N; ct->weight = weight;
        ct->anchor_ids = malloc(sizeof(int) * (size_t)K);
        ct->maps = calloc((size_t)N * (size_t)K, sizeof(int*));
        ko_select_anchors(dist, N, K, ct->anchor_ids);
        for(int i = 0; i < N; i++){
                for(int k = 0; k < K; k++){
                        const int ak = ct->anchor_ids[k];
                        int* map = malloc(sizeof(int) * (size_t)(lens[i] > 0 ? lens[i] : 1));
                        if(i == ak){
                                for(int p = 0; p < lens[i]; p++) map[p] = p;
                        }else{
                                const long long poff = 0;
                                int* coded = malloc(sizeof(int) * (size_t)(lens[i] + lens[ak] + 3));
                                ko_pairwise_batch(codes, off, lens, &i, &ak, 1, subm, gpo, gpe, tgpe, coded, &poff, NULL);
                                ko_posmap(coded, lens[i], map);
                                free(coded);
                        }
                        ct->maps[i * K + k] = map;
                }
        }
        return ct;
}

/*
 * CARRIED VOTES (the device's rule, kalign_amd/csrc/ka_profile.h: ka_votes_merge / ka_cons_from_tables; switched on with
 * ko_set_carried_votes -- the golden tests run every consistency tree with it and expect the reference's answer).
 * get_node_anchor_positions counts, per anchor and column, over the node's members in sip order: best = the position of the first
 * member that has one, total = members that have one, agree = members at best.  sip[c] = rev(sip[a]) ++ rev(sip[b])
 * (aln_run.c:428-436), so c's first voter is a's LAST one.  A node therefore carries both ends of its list per anchor and column --
 * f / fc: first voter's position / voters at it; l / lc: the same for the last voter; n: voters -- and a column of c follows from
 * the columns of a and b it is made of:  one side only (X): f = l_X, fc = lc_X, l = f_X, lc = fc_X, n = n_X;  both: f = l_a,
 * fc = lc_a + #{voters of b at l_a}, l = f_b, lc = fc_b + #{voters of a at f_b}, n = n_a + n_b, where #{voters of X at p} is fc_X
 * if p == f_X, lc_X if p == l_X, 0 if the voters at f_X and l_X are all of X's voters, and otherwise has to be COUNTED over X's
 * members (the device marks the cell and settles all marks with one sweep; here they are counted on the spot and tallied).
 */
typedef struct { int f, fc, l, lc, n; } ko_vote;
static int ko_carried_on = 0;
static long long ko_carried_cells = 0, ko_carried_counted = 0;
static long long ko_carried_distinct[8];
void ko_set_carried_votes(int on) { ko_carried_on = on; ko_carried_cells = 0; ko_carried_counted = 0; for(int i = 0; i < 8; i++) ko_carried_distinct[i] = 0; }
void ko_carried_votes_cells(long long* cells, long long* counted) { *cells = ko_carried_cells; *counted = ko_carried_counted; }

static ko_vote ko_vote_cell(const ko_cons* ct, int node, int nmem, const ko_vote* tab, int plen, int k, int i)
{
        ko_vote v;
        if(nmem == 1){                                             /* a leaf: its position map is its table */
                v.f = v.l = ct->maps[node * ct->K + k][i];
                v.fc = v.lc = v.n = (v.f >= 0) ? 1 : 0;
                return v;
        }
        return tab[(size_t)k * (size_t)plen + (size_t)i];
}

/* voters of node X (members xm[0..xn), alignment in gaps[]) whose position for anchor k in X's column col is p */
static int ko_vote_count(const ko_cons* ct, const int* xm, int xn, const int* lens, int** gaps, int k, int col, int p)
{
        int cnt = 0;
        for(int mi = 0; mi < xn; mi++){
                const int si = xm[mi];
                const int* g = gaps[si];
                int c = 0;
                if(si >= ct->N) continue;
                for(int q = 0; q < lens[si]; q++){
                        c += g[q];
                        if(c == col){ if(ct->maps[si * ct->K + k][q] == p) cnt++; break; }
                        if(c > col) break;
                        c++;
                }
        }
        return cnt;
}

/* statistics for DESIGN section 7 (ko_carried_votes_distinct): how many DIFFERENT positions the voters of a merged cell hold -- what a
   carried cell would have to remember to never need a count */
void ko_carried_votes_distinct(long long* out8) { for(int i = 0; i < 8; i++) out8[i] = ko_carried_distinct[i]; }
static void ko_vote_distinct(const ko_cons* ct, const int* xm, int xn, const int* lens, int** gaps, int k, int col, int* seen, int* nseen)
{
        for(int mi = 0; mi < xn; mi++){
                const int si = xm[mi];
                const int* g = gaps[si];
                int c = 0;
                if(si >= ct->N) continue;
                for(int q = 0; q < lens[si]; q++){
                        c += g[q];
                        if(c == col){
                                const int p = ct->maps[si * ct->K + k][q];
                                int known = p < 0;
                                for(int s = 0; s < *nseen && !known; s++) known = seen[s] == p;
                                if(!known && *nseen < 64) seen[(*nseen)++] = p;
                                break;
                        }
                        if(c > col) break;
                        c++;
                }
        }
}

static int ko_vote_known(const ko_vote* x, int p, int* cnt)
{
        if(p == x->f){ *cnt = x->fc; return 1; }
        if(p == x->l){ *cnt = x->lc; return 1; }
        if((x->f == x->l) ? (x->fc == x->n) : (x->fc + x->lc == x->n)){ *cnt = 0; return 1; }
        return 0;
}

/* the table of c = merge(a, b) along `coded`; BEFORE weave: gaps[] still hold every member's columns within a and within b */
static ko_vote* ko_votes_merge(const ko_cons* ct, const int* coded, int a, int na, const int* ma, const ko_vote* ta, int la,
                               int b, int nb, const int* mb, const ko_vote* tb, int lb, const int* lens, int** gaps)
{
        const int K = ct->K, alnlen = coded[0];
        ko_vote* vt = malloc(sizeof(ko_vote) * (size_t)K * (size_t)alnlen);
        int ia = 0, ib = 0;
        for(int j = 0; j < alnlen; j++){
                const int code = coded[j + 1];
                const int ca = (code & 1) ? -1 : ia, cb = (!(code & 1) && (code & 2)) ? -1 : ib;
                for(int k = 0; k < K; k++){
                        ko_vote A = { -1, 0, -1, 0, 0 }, B = { -1, 0, -1, 0, 0 }, v;
                        int cnt;
                        if(ca >= 0) A = ko_vote_cell(ct, a, na, ta, la, k, ca);
                        if(cb >= 0) B = ko_vote_cell(ct, b, nb, tb, lb, k, cb);
                        if(A.n == 0){ v.f = B.l; v.fc = B.lc; v.l = B.f; v.lc = B.fc; v.n = B.n; }
                        else if(B.n == 0){ v.f = A.l; v.fc = A.lc; v.l = A.f; v.lc = A.fc; v.n = A.n; }
                        else{
                                v.f = A.l; v.l = B.f; v.n = A.n + B.n;
                                if(!ko_vote_known(&B, v.f, &cnt)){ cnt = ko_vote_count(ct, mb, nb, lens, gaps, k, cb, v.f); ko_carried_counted++; }
                                v.fc = A.lc + cnt;
                                if(!ko_vote_known(&A, v.l, &cnt)){ cnt = ko_vote_count(ct, ma, na, lens, gaps, k, ca, v.l); ko_carried_counted++; }
                                v.lc = B.fc + cnt;
                        }
                        vt[(size_t)k * (size_t)alnlen + (size_t)j] = v;
                        ko_carried_cells++;
                        if(ko_carried_on > 1 && A.n > 0 && B.n > 0){           /* (ko_set_carried_votes(2): with the statistics) */
                                int seen[64], nseen = 0;
                                ko_vote_distinct(ct, ma, na, lens, gaps, k, ca, seen, &nseen);
                                ko_vote_distinct(ct, mb, nb, lens, gaps, k, cb, seen, &nseen);
                                ko_carried_distinct[nseen < 7 ? nseen : 7]++;
                        }
                }
                if(ca >= 0) ia++;
                if(cb >= 0) ib++;
        }
        return vt;
}

/* get_node_anchor_positions (anchor_consistency.c:352-470); tab != NULL: from the node's carried table instead of a count */
static void ko_node_positions(const ko_cons* ct, int node, int nmem, const int* members, const int* lens,
                              int** gaps, int dp_len, int k, int* positions, float* conf, const ko_vote* tab)
{
        const int K = ct->K;
        if(tab && nmem > 1){
                for(int i = 0; i < dp_len; i++){
                        const ko_vote v = tab[(size_t)k * (size_t)dp_len + (size_t)i];
                        if(v.n > 0 && v.fc > 0){ positions[i] = v.f; conf[i] = (float)v.fc / (float)v.n; }
                        else { positions[i] = -1; conf[i] = 0.0f; }
                }
                return;
        }
        if(nmem == 1){
                const int* map = ct->maps[node * K + k];
                const int seq_len = lens[node];
                int i;
                for(i = 0; i < dp_len && i < seq_len; i++){
                        positions[i] = map[i];
                        conf[i] = (map[i] >= 0) ? 1.0f : 0.0f;
                }
                for(; i < dp_len; i++){ positions[i] = -1; conf[i] = 0.0f; }
                return;
        }
        {
                int* c2u = malloc(sizeof(int) * (size_t)(dp_len + 1));
                int* best = malloc(sizeof(int) * (size_t)dp_len);
                int* agree = calloc((size_t)dp_len, sizeof(int));
                int* total = calloc((size_t)dp_len, sizeof(int));
                for(int c = 0; c < dp_len; c++) best[c] = -1;
                for(int mi = 0; mi < nmem; mi++){
                        const int si = members[mi];
                        const int* map;
                        const int* g;
                        int seq_len, col = 0;
                        if(si >= ct->N) continue;
                        map = ct->maps[si * K + k];
                        seq_len = lens[si];
                        g = gaps[si];
                        for(int p = 0; p <= seq_len && col < dp_len; p++){
                                for(int q = 0; q < g[p] && col < dp_len; q++) c2u[col++] = -1;
                                if(p < seq_len && col < dp_len) c2u[col++] = p;
                        }
                        while(col < dp_len) c2u[col++] = -1;
                        for(int c = 0; c < dp_len; c++){
                                const int ugp = c2u[c];
                                int apos;
                                if(ugp < 0 || ugp >= seq_len) continue;
                                apos = map[ugp];
                                if(apos < 0) continue;
                                total[c]++;
                                if(best[c] < 0){ best[c] = apos; agree[c] = 1; }
                                else if(apos == best[c]) agree[c]++;
                        }
                }
                for(int c = 0; c < dp_len; c++){
                        if(total[c] > 0 && agree[c] > 0){
                                positions[c] = best[c];
                                conf[c] = (float)agree[c] / (float)total[c];
                        }else{
                                positions[c] = -1; conf[c] = 0.0f;
                        }
                }
                free(c2u); free(best); free(agree); free(total);
        }
}

/* anchor_consistency_get_bonus_profile (anchor_consistency.c:472-561): dense rows x cols matrix */
static float* ko_bonus_profile(const ko_cons* ct, const int* lens, int** gaps,
                               int rnode, int rn, const int* rmem, int rows,
                               int cnode, int cn, const int* cmem, int cols, const ko_vote* rtab, const ko_vote* ctab)
{
        float* bonus = calloc((size_t)rows * (size_t)cols, sizeof(float));
        int* apos_a = malloc(sizeof(int) * (size_t)rows);
        float* conf_a = malloc(sizeof(float) * (size_t)rows);
        int* apos_b = malloc(sizeof(int) * (size_t)cols);
        float* conf_b = malloc(sizeof(float) * (size_t)cols);
        const float paw = ct->weight / (float)ct->K;
        for(int k = 0; k < ct->K; k++){
                int anchor_len = 0;
                int* inv_b;
                float* inv_conf_b;
                ko_node_positions(ct, rnode, rn, rmem, lens, gaps, rows, k, apos_a, conf_a, rtab);
                ko_node_positions(ct, cnode, cn, cmem, lens, gaps, cols, k, apos_b, conf_b, ctab);
                for(int i = 0; i < rows; i++) if(apos_a[i] >= anchor_len) anchor_len = apos_a[i] + 1;
                for(int j = 0; j < cols; j++) if(apos_b[j] >= anchor_len) anchor_len = apos_b[j] + 1;
                if(anchor_len == 0) continue;
                inv_b = malloc(sizeof(int) * (size_t)anchor_len);
                inv_conf_b = malloc(sizeof(float) * (size_t)anchor_len);
                for(int j = 0; j < anchor_len; j++){ inv_b[j] = -1; inv_conf_b[j] = 0.0f; }
                for(int j = 0; j < cols; j++){
                        if(apos_b[j] >= 0 && apos_b[j] < anchor_len){
                                inv_b[apos_b[j]] = j;
                                inv_conf_b[apos_b[j]] = conf_b[j];
                        }
                }
                for(int i = 0; i < rows; i++){
                        const int ak = apos_a[i];
                        if(ak >= 0 && ak < anchor_len){
                                const int bj = inv_b[ak];
                                if(bj >= 0) bonus[(size_t)i * (size_t)cols + (size_t)bj] += paw * conf_a[i] * inv_conf_b[ak];
                        }
                }
                free(inv_b); free(inv_conf_b);
        }
        free(apos_a); free(conf_a); free(apos_b); free(conf_b);
        return bonus;
}

/*
 * The dispatcher: create_msa_tree / do_align (aln_run.c:43-124, :213-441), with the
 * anchor-consistency bonus when n_anchors > 0 (aln_wrap.c:207-214, aln_run.c:262-295).
 * Tasks must be in TASK_ORDER_TREE order (children before parents; the last task is the root).
 */
/* refine_mode 0: create_msa_tree.  3: create_msa_tree_inline_refine (see below).  1 / 2: refine_alignment (aln_refine.c:36-88; KALIGN_REFINE_ALL / _CONFIDENT) --
   the same walk over the edges with convert_raw_path coding, and refine_edge's five trials (:93-346) on the edges to
   refine; conf_in = the first pass's task confidences (mode 2: edges at or below their median are refined). */
static int ko_tree_impl(int numseq, const uint8_t* codes, const int* off, const int* lens,
                     const float* seq_distances,
                     int n_tasks, const int* abc,
                     const float* subm, const float* scal,
                     int n_anchors, float cons_weight,
                     ko_task_rec* recs, int* paths_out, long long paths_cap,
                     int* gaps_out, int dump_task, float* prof_dump,
                     int* anchor_ids_out, int* maps_out, uint64_t* bonus_hash_out,
                     int refine_mode_in, const float* conf_in)
{
        const int refine_mode = refine_mode_in & 255;
        const int adaptive = (refine_mode_in >> 8) & 1;             /* + 256: ap->adaptive_budget */
        const int nprof = 2 * numseq - 1;
        const float gpo0 = scal[0], gpe0 = scal[1], tgpe0 = scal[2];
        const float dist_scale = scal[3], vsm_amax = scal[4], usw = scal[5];
        float** prof = calloc((size_t)nprof, sizeof(float*));
        int** sip = calloc((size_t)nprof, sizeof(int*));
        int* nsip = calloc((size_t)nprof, sizeof(int));
        int* plen = calloc((size_t)nprof, sizeof(int));
        int** gaps = calloc((size_t)numseq, sizeof(int*));
        ko_vote** votes = NULL;                                    /* carried vote tables per node (ko_set_carried_votes) */
        long long poff = 0;
        int rc = 0;
        float conf_threshold = 0.0f;
        ko_cons* ct = ko_cons_build(numseq, codes, off, lens, seq_distances, subm, gpo0, gpe0, tgpe0, n_anchors, cons_weight);

        if(refine_mode == 2){                                      /* compute_confidence_threshold, aln_refine.c:674-712 */
                float* cf = malloc(sizeof(float) * (size_t)n_tasks);
                for(int i = 0; i < n_tasks; i++) cf[i] = conf_in[i];
                for(int i = 1; i < n_tasks; i++){
                        float tmp = cf[i];
                        int j = i - 1;
                        while(j >= 0 && cf[j] > tmp){ cf[j + 1] = cf[j]; j--; }
                        cf[j + 1] = tmp;
                }
                conf_threshold = (n_tasks % 2 == 0) ? (cf[n_tasks / 2 - 1] + cf[n_tasks / 2]) / 2.0F : cf[n_tasks / 2];
                free(cf);
        }

        if(ct && ko_carried_on) votes = calloc((size_t)nprof, sizeof(ko_vote*));
        if(ct && anchor_ids_out) for(int k = 0; k < ct->K; k++) anchor_ids_out[k] = ct->anchor_ids[k];
        if(ct && maps_out){
                int o = 0;
                for(int i = 0; i < numseq * ct->K; i++) for(int p = 0; p < lens[i / ct->K]; p++) maps_out[o++] = ct->maps[i][p];
        }
        for(int i = 0; i < numseq; i++){
                sip[i] = malloc(sizeof(int)); sip[i][0] = i; nsip[i] = 1;
                gaps[i] = calloc((size_t)lens[i] + 1, sizeof(int));
        }
        for(int tid = 0; tid < n_tasks && !rc; tid++){
                const int a = abc[3 * tid], b = abc[3 * tid + 1], c = abc[3 * tid + 2];
                ko_task_rec* r = &recs[tid];
                float gap_scale = 1.0f, soff = 0.0f, gpo = gpo0, gpe = gpe0, tgpe = tgpe0;
                int len_a, len_b, cnt, g, swapped = 0;
                dp_t d;
                probe_t probe;
                int *raw, *raw2, *coded;
                float* merged;
                float* bonus = NULL;

                memset(&d, 0, sizeof(d));
                memset(&probe, 0, sizeof(probe));
                if(dist_scale > 0.0f && seq_distances){                   /* compute_gap_scale */
                        float avg = mean_distance(seq_distances, sip[a], nsip[a], sip[b], nsip[b], numseq, &cnt);
                        if(cnt){
                                gap_scale = 1.0f - dist_scale * avg;
                                if(gap_scale < 0.3f) gap_scale = 0.3f;
                                if(gap_scale > 1.0f) gap_scale = 1.0f;
                        }
                }
                if(vsm_amax > 0.0f && seq_distances){                      /* compute_subm_offset */
                        float avg = mean_distance(seq_distances, sip[a], nsip[a], sip[b], nsip[b], numseq, &cnt);
                        if(cnt){
                                soff = vsm_amax - avg;
                                if(soff < 0.0f) soff = 0.0f;
                        }
                }
                if(gap_scale < 1.0f || soff > 0.0f){                       /* aln_run.c:229-237 */
                        gpo *= gap_scale; gpe *= gap_scale; tgpe *= gap_scale;
                }else{
                        soff = 0.0f;
                }
                if(nsip[a] == 1){
                        len_a = lens[a];
                        prof[a] = malloc(sizeof(float) * 64 * (size_t)(len_a + 2));
                        ko_make_profile(codes + off[a], len_a, subm, gpo, gpe, tgpe, soff, prof[a]);
                }else{
                        len_a = plen[a];
                        ko_set_gap_penalties(prof[a], len_a, nsip[b]);
                }
                if(nsip[b] == 1){
                        len_b = lens[b];
                        prof[b] = malloc(sizeof(float) * 64 * (size_t)(len_b + 2));
                        ko_make_profile(codes + off[b], len_b, subm, gpo, gpe, tgpe, soff, prof[b]);
                }else{
                        len_b = plen[b];
                        ko_set_gap_penalties(prof[b], len_b, nsip[a]);
                }

                /* operand selection / swap (aln_run.c:297-388) */
                d.subm = subm; d.gpo = gpo; d.gpe = gpe; d.tgpe = tgpe; d.soff = soff;
                if(nsip[a] == 1 && nsip[b] == 1){
                        d.kind = K_SS;
                        if(len_a < len_b){ d.s1 = codes + off[a]; d.s2 = codes + off[b]; }
                        else{ swapped = 1; d.s1 = codes + off[b]; d.s2 = codes + off[a]; }
                }else if(nsip[a] == 1){
                        d.kind = K_SP; swapped = 1;
                        d.s2 = codes + off[a]; d.p1 = prof[b];
                        d.sp_open = gpo * (float)nsip[b]; d.sp_ext = gpe * (float)nsip[b]; d.sp_text = tgpe * (float)nsip[b];
                }else if(nsip[b] == 1){
                        d.kind = K_SP;
                        d.s2 = codes + off[b]; d.p1 = prof[a];
                        d.sp_open = gpo * (float)nsip[a]; d.sp_ext = gpe * (float)nsip[a]; d.sp_text = tgpe * (float)nsip[a];
                }else{
                        d.kind = K_PP;
                        if(len_a < len_b){ d.p1 = prof[a]; d.p2 = prof[b]; }
                        else{ swapped = 1; d.p1 = prof[b]; d.p2 = prof[a]; }
                }
                d.len_a = swapped ? len_b : len_a;
                d.len_b = swapped ? len_a : len_b;
                if(ct){
                        /* rows/cols of the bonus follow the DP operands (aln_run.c:262-295) */
                        const int rn = swapped ? b : a, cn = swapped ? a : b;
                        bonus = ko_bonus_profile(ct, lens, gaps, rn, nsip[rn], sip[rn], d.len_a, cn, nsip[cn], sip[cn], d.len_b,
                                                 votes ? votes[rn] : NULL, votes ? votes[cn] : NULL);
                        d.bonus = bonus; d.bstride = d.len_b;
                        if(bonus_hash_out) bonus_hash_out[tid] = ko_fnv1a(bonus, sizeof(float) * (uint64_t)d.len_a * (uint64_t)d.len_b);
                }else if(bonus_hash_out){
                        bonus_hash_out[tid] = 0;
                }
                g = (len_a > len_b ? len_a : len_b) + 2;
                d.f = malloc(sizeof(st) * (size_t)g);
                d.b = malloc(sizeof(st) * (size_t)g);
                raw = malloc(sizeof(int) * (size_t)(len_a + len_b + 2));
                raw2 = malloc(sizeof(int) * (size_t)(len_a + len_b + 2));
                coded = malloc(sizeof(int) * (size_t)(len_a + len_b + 3));
                d.path = raw;
                float task_conf = 0.0f;
                if(!refine_mode){
                        ko_align(&d, &probe);
                        if(swapped){
                                ko_mirror_path(raw, len_a, len_b, raw2);
                                ko_code_path(raw2, len_a, len_b, coded);
                        }else{
                                ko_code_path(raw, len_a, len_b, coded);
                        }
                        task_conf = d.mcount > 0 ? d.msum / (float)d.mcount : 0.0f;
                }else{
                        /* refine_edge (multi-trial) or replay_edge (one trial); trial 0 is the deterministic baseline,
                           trials 1..4 flip uncertain meetups round-robin with the baseline's mean margin as threshold */
                        /* refine_mode 3 = KALIGN_REFINE_INLINE (do_align_inline_refine, aln_run.c:515-790, called with three
                           trials by aln_wrap.c:222-224): every edge, first-pass path coding, confidence = the best SP score */
                        const int inline_mode = refine_mode == 3;
                        const int refine_it = refine_mode == 1 || inline_mode || (refine_mode == 2 && conf_in[tid] <= conf_threshold);
                        int n_trials = inline_mode ? 3 : refine_it ? 5 : 1;
                        /* adaptive budget (aln_refine.c:187-193, 255-282; refine_edge only): the baseline's margins are kept and
                           the number of trials follows from the share of very uncertain meetups */
                        const int adaptive_it = adaptive && refine_it && !inline_mode;
                        if(adaptive_it){
                                int est = (len_a < len_b ? len_a : len_b) + 1;
                                if(est < 64) est = 64;
                                d.flip_margins = malloc(sizeof(float) * (size_t)est);
                                d.flip_margin_alloc = est;
                        }
                        int* cand = malloc(sizeof(int) * (size_t)(len_a + len_b + 3));
                        float best_sp = -F, avg_margin = 0.0F, best_msum = 0.0F;
                        int best_mcount = 0;
                        for(int k = 0; k < n_trials; k++){
                                probe_t skip;
                                memset(&skip, 0, sizeof(skip));
                                skip.have_top = 1;
                                d.flip_threshold = (k == 0) ? 0.0F : avg_margin;
                                d.flip_trial = k; d.flip_stride = n_trials - 1; d.flip_counter = 0;
                                ko_align(&d, k == 0 ? &probe : &skip);
                                if(swapped) ko_mirror_path(raw, len_a, len_b, raw2);
                                if(inline_mode) ko_code_path(swapped ? raw2 : raw, len_a, len_b, cand);        /* add_gap_info_to_path_n, aln_run.c:713 */
                                else ko_convert_raw_path(swapped ? raw2 : raw, len_a, len_b, cand);
                                if(refine_it){
                                        const float sp = ko_sp_score(codes, off, lens, gaps, cand, sip[a], nsip[a], sip[b], nsip[b], subm, gpo, gpe, tgpe);
                                        if(sp > best_sp){
                                                best_sp = sp; best_msum = d.msum; best_mcount = d.mcount;
                                                memcpy(coded, cand, sizeof(int) * (size_t)(cand[0] + 2));
                                        }
                                }else{
                                        best_msum = d.msum; best_mcount = d.mcount;
                                        memcpy(coded, cand, sizeof(int) * (size_t)(cand[0] + 2));
                                }
                                if(k == 0 && d.mcount > 0) avg_margin = d.msum / (float)d.mcount;
                                if(k == 0 && adaptive_it){
                                        if(d.mcount > 0){
                                                int n_vu = 0;
                                                const float vu = avg_margin * 0.25F;
                                                const int seen = d.mcount < d.flip_margin_alloc ? d.mcount : d.flip_margin_alloc;   /* (beyond it the reference reads past its buffer) */
                                                for(int m_i = 0; m_i < seen; m_i++) if(d.flip_margins[m_i] < vu) n_vu++;
                                                const float frac = (float)n_vu / (float)d.mcount;
                                                n_trials = 1 + (int)(7.0F * frac + 0.5F);
                                        }
                                        free(d.flip_margins); d.flip_margins = NULL;
                                }
                        }
                        d.flip_threshold = 0.0F;
                        task_conf = best_mcount > 0 ? best_msum / (float)best_mcount : 0.0f;
                        if(inline_mode) task_conf = best_sp;                   /* aln_run.c:742 */
                        free(cand);
                }

                r->a = a; r->b = b; r->c = c;
                r->len_a = len_a; r->len_b = len_b; r->nsip_a = nsip[a]; r->nsip_b = nsip[b];
                r->plen = coded[0]; r->kind = d.kind; r->swapped = swapped;
                r->meet = probe.have_top ? probe.top_meet : -1;
                r->transition = probe.have_top ? probe.top_tr : -1;
                r->score = probe.have_top ? probe.top_score : 0.0f;
                r->fhash = probe.fhash; r->bhash = probe.bhash;
                r->gap_scale = gap_scale; r->subm_off = soff;
                r->confidence = task_conf;
                r->prof_hash = 0;
                if(poff + coded[0] + 2 > paths_cap){
                        rc = 2;
                }else{
                        r->path_off = (int)poff;
                        memcpy(paths_out + poff, coded, sizeof(int) * (size_t)(coded[0] + 2));
                        poff += coded[0] + 2;
                }

                /* merged profile with the UNSCALED penalties (aln_run.c:405-413) */
                merged = malloc(sizeof(float) * 64 * (size_t)(coded[0] + 2));
                if(tid != n_tasks - 1){
                        ko_update_profile(prof[a], prof[b], merged, coded, nsip[a], nsip[b], subm, gpo0, gpe0, tgpe0, usw);
                        r->prof_hash = ko_fnv1a(merged, sizeof(float) * 64 * (uint64_t)(coded[0] + 2));
                        if(tid == dump_task && prof_dump){
                                memcpy(prof_dump, merged, sizeof(float) * 64 * (size_t)(coded[0] + 2));
                        }
                }
                free(prof[a]); free(prof[b]); prof[a] = NULL; prof[b] = NULL;
                prof[c] = merged;
                if(votes && tid != n_tasks - 1)
                        votes[c] = ko_votes_merge(ct, coded, a, nsip[a], sip[a], votes[a], len_a, b, nsip[b], sip[b], votes[b], len_b, lens, gaps);
                if(votes){ free(votes[a]); free(votes[b]); votes[a] = NULL; votes[b] = NULL; }
                weave(coded, sip[a], nsip[a], sip[b], nsip[b], lens, gaps);
                plen[c] = coded[0];
                nsip[c] = nsip[a] + nsip[b];
                sip[c] = malloc(sizeof(int) * (size_t)nsip[c]);
                g = 0;
                for(int j = nsip[a]; j--;) sip[c][g++] = sip[a][j];          /* aln_run.c:428-436 */
                for(int j = nsip[b]; j--;) sip[c][g++] = sip[b][j];
                free(d.f); free(d.b); free(raw); free(raw2); free(coded); free(bonus);
        }
        ko_cons_free(ct);
        if(gaps_out){
                int o = 0;
                for(int i = 0; i < numseq; i++) for(int j = 0; j <= lens[i]; j++) gaps_out[o++] = gaps[i][j];
        }
        for(int i = 0; i < nprof; i++){ free(prof[i]); free(sip[i]); if(votes) free(votes[i]); }
        for(int i = 0; i < numseq; i++) free(gaps[i]);
        free(prof); free(sip); free(nsip); free(plen); free(gaps); free(votes);
        return rc;
}

int ko_msa_tree_cons(int numseq, const uint8_t* codes, const int* off, const int* lens,
                     const float* seq_distances,
                     int n_tasks, const int* abc,
                     const float* subm, const float* scal,
                     int n_anchors, float cons_weight,
                     ko_task_rec* recs, int* paths_out, long long paths_cap,
                     int* gaps_out, int dump_task, float* prof_dump,
                     int* anchor_ids_out, int* maps_out, uint64_t* bonus_hash_out)
{
        return ko_tree_impl(numseq, codes, off, lens, seq_distances, n_tasks, abc, subm, scal, n_anchors, cons_weight,
                            recs, paths_out, paths_cap, gaps_out, dump_task, prof_dump, anchor_ids_out, maps_out, bonus_hash_out, 0, NULL);
}

/* refine_alignment (aln_refine.c:36-88) as its own pass: mode 1 = KALIGN_REFINE_ALL, 2 = KALIGN_REFINE_CONFIDENT with
   conf_in[n_tasks] = the task confidences of the alignment being refined.  recs[t].confidence / plen / paths are those of
   the refined edges (paths in convert_raw_path coding), gaps_out the refined alignment. */
int ko_msa_tree_refine(int numseq, const uint8_t* codes, const int* off, const int* lens,
                       const float* seq_distances,
                       int n_tasks, const int* abc,
                       const float* subm, const float* scal,
                       int n_anchors, float cons_weight, int mode, const float* conf_in,
                       ko_task_rec* recs, int* paths_out, long long paths_cap, int* gaps_out)
{
        if((mode & 255) < 1 || (mode & 255) > 3 || (mode & ~0x1ff)) return 1;      /* + 256: adaptive budget (modes 1, 2) */
        if((mode & 255) == 2 && !conf_in) return 1;
        return ko_tree_impl(numseq, codes, off, lens, seq_distances, n_tasks, abc, subm, scal, n_anchors, cons_weight,
                            recs, paths_out, paths_cap, gaps_out, -1, NULL, NULL, NULL, NULL, mode, conf_in);
}

int ko_msa_tree(int numseq, const uint8_t* codes, const int* off, const int* lens,
                const float* seq_distances,
                int n_tasks, const int* abc,
                const float* subm, const float* scal,
                ko_task_rec* recs, int* paths_out, long long paths_cap,
                int* gaps_out, int dump_task, float* prof_dump)
{
        return ko_msa_tree_cons(numseq, codes, off, lens, seq_distances, n_tasks, abc, subm, scal, 0, 0.0f,
                                recs, paths_out, paths_cap, gaps_out, dump_task, prof_dump, NULL, NULL, NULL);
}

/* N x seq-seq as pairwise_align_map runs them (anchor_consistency.c:19-120) */
int ko_pairwise_batch(const uint8_t* codes, const int* off, const int* lens,
                      const int* ia, const int* ib, int npairs,
                      const float* subm, float gpo, float gpe, float tgpe,
                      int* paths_out, const long long* poff, float* scores_out)
{
        for(int k = 0; k < npairs; k++){
                const int i = ia[k], j = ib[k];
                const int len_i = lens[i], len_j = lens[j];
                const int swapped = !(len_i <= len_j);
                const int g = (len_i > len_j ? len_i : len_j) + 2;
                dp_t d;
                probe_t probe;
                int* raw = malloc(sizeof(int) * (size_t)(len_i + len_j + 2));
                int* raw2 = malloc(sizeof(int) * (size_t)(len_i + len_j + 2));
                memset(&d, 0, sizeof(d));
                memset(&probe, 0, sizeof(probe));
                d.kind = K_SS; d.subm = subm; d.gpo = gpo; d.gpe = gpe; d.tgpe = tgpe; d.soff = 0.0f;
                d.s1 = codes + off[swapped ? j : i]; d.s2 = codes + off[swapped ? i : j];
                d.len_a = swapped ? len_j : len_i; d.len_b = swapped ? len_i : len_j;
                d.f = malloc(sizeof(st) * (size_t)g); d.b = malloc(sizeof(st) * (size_t)g);
                d.path = raw;
                ko_align(&d, &probe);
                if(swapped){
                        ko_mirror_path(raw, len_i, len_j, raw2);
                        ko_code_path(raw2, len_i, len_j, paths_out + poff[k]);
                }else{
                        ko_code_path(raw, len_i, len_j, paths_out + poff[k]);
                }
                if(scores_out) scores_out[k] = probe.have_top ? probe.top_score : 0.0f;
                free(d.f); free(d.b); free(raw); free(raw2);
        }
        return 0;
}

int ko_dp_single(int kind, const uint8_t* seq1, const uint8_t* seq2,
                 const float* prof1, const float* prof2, int len_a, int len_b,
                 const float* subm, float gpo, float gpe, float tgpe, float soff, int sip,
                 const float* bonus, int bonus_stride,
                 int* raw_path, float* f_out, float* b_out,
                 int* meet, int* transition, float* score, float* confidence)
{
        dp_t d;
        probe_t probe;
        const int g = (len_a > len_b ? len_a : len_b) + 2;
        memset(&d, 0, sizeof(d));
        memset(&probe, 0, sizeof(probe));
        d.kind = kind; d.s1 = seq1; d.s2 = seq2; d.p1 = prof1; d.p2 = prof2;
        d.len_a = len_a; d.len_b = len_b; d.subm = subm;
        d.gpo = gpo; d.gpe = gpe; d.tgpe = tgpe; d.soff = soff;
        d.sp_open = gpo * (float)sip; d.sp_ext = gpe * (float)sip; d.sp_text = tgpe * (float)sip;
        d.bonus = bonus; d.bstride = bonus_stride;
        d.f = malloc(sizeof(st) * (size_t)g); d.b = malloc(sizeof(st) * (size_t)g);
        d.path = malloc(sizeof(int) * (size_t)(len_a + len_b + 2));
        probe.f_out = f_out; probe.b_out = b_out;
        ko_align(&d, &probe);
        memcpy(raw_path, d.path, sizeof(int) * (size_t)(len_a + 2));
        if(meet) *meet = probe.have_top ? probe.top_meet : -1;
        if(transition) *transition = probe.have_top ? probe.top_tr : -1;
        if(score) *score = probe.have_top ? probe.top_score : 0.0f;
        if(confidence) *confidence = d.mcount > 0 ? d.msum / (float)d.mcount : 0.0f;
        free(d.f); free(d.b); free(d.path);
        return 0;
}

/* ---- distance estimation: block-wise Myers bit-vector edit distance (bpm_block, lib/src/bpm.c:356-582) ----
 * Semi-global: the pattern p (length m, at most 1024 positions are used) against every end position of the
 * text t; 64-bit blocks with Ukkonen's band: y = last active block.  Alphabet codes < 13 (bpm.c:11).
 * Restated with the reference's exact update order, including its band rules: the band starts with ALL
 * blocks active (maxd = m), shrinks while the last block's score is >= m + 64 and re-opens a block with
 * P = ~0, M = 0 (bpm.c:505-560).  Text positions n .. n+W-1 are padding with code 0 (W = 64*b_max - m). */
int ko_bpm_block(const uint8_t* t, const uint8_t* p, int n, int m)
{
        enum { W64 = 64, SIG = 13 };
        uint64_t peq[13][16];
        uint64_t P[16], M[16];
        int32_t score[16];
        const uint64_t ONE = 1, HIGH = ONE << 63;
        int b_max, Wpad, k, maxd, y;
        if(m > 1024) m = 1024;
        b_max = (m == 0) ? 1 : (m / W64 + ((m % W64) ? 1 : 0));
        Wpad = W64 * b_max - m;
        k = m; maxd = m;
        memset(peq, 0, sizeof(peq));
        for(int c = 0; c < SIG; c++){
                for(int b = 0; b < b_max; b++){
                        uint64_t bit = 1;
                        for(int i = b * W64; i < (b + 1) * W64; i++){
                                if(i >= m || p[i] == c) peq[c][b] |= bit;      /* beyond m: matches anything */
                                bit <<= 1;
                        }
                }
        }
        y = ((maxd == 0) ? 1 : (maxd / W64 + ((maxd % W64) ? 1 : 0))) - 1;
        for(int b = 0; b < 16; b++){ P[b] = 0; M[b] = 0; score[b] = 0; }
        for(int b = 0; b <= y; b++){ P[b] = ~(uint64_t)0; M[b] = 0; score[b] = (b + 1) * W64; }
        for(int i = 0; i < n + Wpad; i++){
                const int c = (i >= n) ? 0 : t[i];
                int carry = 0;
                for(int b = 0; b <= y; b++){
                        uint64_t Pv = P[b], Mv = M[b], Eq = peq[c][b], Xv, Xh, Ph, Mh;
                        const int hin = carry;
                        int hout = 0;
                        Xv = Eq | Mv;
                        if(hin < 0) Eq |= ONE;
                        Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                        Ph = Mv | ~(Xh | Pv);
                        Mh = Pv & Xh;
                        if(Ph & HIGH) hout += 1;
                        if(Mh & HIGH) hout -= 1;
                        Ph <<= 1; Mh <<= 1;
                        if(hin < 0) Mh |= ONE; else if(hin > 0) Ph |= ONE;
                        P[b] = Mh | ~(Xv | Ph);
                        M[b] = Ph & Xv;
                        carry = hout;
                        score[b] += carry;
                }
                if((score[y] - carry <= maxd) && (y < b_max - 1) && ((peq[c][y + 1] & ONE) || (carry < 0))){
                        uint64_t Pv, Mv, Eq, Xv, Xh, Ph, Mh;
                        const int hin = carry;
                        int hout = 0;
                        y += 1;
                        Pv = ~(uint64_t)0; Mv = 0; Eq = peq[c][y];
                        Xv = Eq | Mv;
                        if(hin < 0) Eq |= ONE;
                        Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                        Ph = Mv | ~(Xh | Pv);
                        Mh = Pv & Xh;
                        if(Ph & HIGH) hout += 1;
                        if(Mh & HIGH) hout -= 1;
                        Ph <<= 1; Mh <<= 1;
                        if(hin < 0) Mh |= ONE; else if(hin > 0) Ph |= ONE;
                        P[y] = Mh | ~(Xv | Ph);
                        M[y] = Ph & Xv;
                        score[y] = score[y - 1] + W64 - carry + hout;
                }else{
                        while(score[y] >= maxd + W64){
                                if(y == 0) break;
                                y -= 1;
                        }
                }
                if(score[y] < k) k = score[y];
        }
        return k;
}

/* calc_distance (sequence_distance.c:150-162) for a list of pairs: the longer sequence is the text */
int ko_bpm_batch(const uint8_t* codes, const int* off, const int* lens, const int* ia, const int* ib, int npairs, int* dist_out)
{
        for(int k = 0; k < npairs; k++){
                const int a = ia[k], b = ib[k];
                if(lens[a] > lens[b]) dist_out[k] = ko_bpm_block(codes + off[a], codes + off[b], lens[a], lens[b]);
                else dist_out[k] = ko_bpm_block(codes + off[b], codes + off[a], lens[b], lens[a]);
        }
        return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Realignment pass (kalign_run_realign, lib/src/aln_wrap.c:449-495): the guide tree derived from a
 * finished alignment.  Restated for the tests of ka_aln_guide_tree; pinned against
 * tests/golden/realign_*.npz (made by the reference's own loop).
 * ------------------------------------------------------------------------------------------------ */

/* compute_aln_pairwise_dist (lib/src/aln_apair_dist.c:9-86): dm[i][j] = 1 - matches / aligned over the
   columns where both rows carry a residue; 1 when there is no such column; 0 on the diagonal. */
int ko_aln_pairwise_dist(const uint8_t* rows, int n, long long stride, int alnlen, uint8_t gap, float* dm)
{
        for (int i = 0; i < n; i++) {
                dm[(size_t)i * n + i] = 0.0f;
                for (int j = i + 1; j < n; j++) {
                        const uint8_t* x = rows + (size_t)i * stride;
                        const uint8_t* y = rows + (size_t)j * stride;
                        int both = 0, same = 0;
                        for (int c = 0; c < alnlen; c++) {
                                if (x[c] == gap || y[c] == gap) continue;
                                both++;
                                same += x[c] == y[c];
                        }
                        const float d = both ? 1.0f - (float)same / (float)both : 1.0f;
                        dm[(size_t)i * n + j] = d;
                        dm[(size_t)j * n + i] = d;
                }
        }
        return 0;
}

/* build_tree_from_pairwise (lib/src/bisectingKmeans.c:1150-1200): per-sequence mean distance to the others
   (summed in index order), then upgma (:974-1053) -- repeatedly the first smallest dm[i][j], i < j, among the
   slots still alive; slot i takes over, its row and column become (old_i + old_j) / 2 + 0.001 -- then
   label_internal / create_tasks / sort_tasks: internal nodes numbered in post-order from n, one task per
   internal node, ordered by c.  dm is consumed.  tasks_abc receives 3 * (n - 1) ints. */
int ko_tree_from_pairwise(float* dm, int n, int* tasks_abc, float* seq_distances)
{
        if (n < 2) return 1;
        if (seq_distances) {
                for (int i = 0; i < n; i++) {
                        float sum = 0.0f;
                        for (int j = 0; j < n; j++) if (j != i) sum += dm[(size_t)i * n + j];
                        seq_distances[i] = sum / (float)(n - 1);
                }
        }
        /* tree nodes: 0..n-1 leaves, n.. internal in creation order */
        int* left = malloc(sizeof(int) * 2 * (size_t)n);
        int* right = malloc(sizeof(int) * 2 * (size_t)n);
        int* slot = malloc(sizeof(int) * (size_t)n);      /* node currently held by matrix slot i, -1: merged away */
        int* label = malloc(sizeof(int) * 2 * (size_t)n);
        int* stack = malloc(sizeof(int) * 4 * (size_t)n);
        if (!left || !right || !slot || !label || !stack) { free(left); free(right); free(slot); free(label); free(stack); return 1; }
        for (int i = 0; i < n; i++) { slot[i] = i; left[i] = right[i] = -1; }
        int next = n, keep = 0, drop = 0;
        for (int merge = 0; merge < n - 1; merge++) {
                float best = FLT_MAX;
                for (int i = 0; i < n - 1; i++) {
                        if (slot[i] < 0) continue;
                        for (int j = i + 1; j < n; j++)
                                if (slot[j] >= 0 && dm[(size_t)i * n + j] < best) { best = dm[(size_t)i * n + j]; keep = i; drop = j; }
                }
                left[next] = slot[keep]; right[next] = slot[drop];
                slot[keep] = next++; slot[drop] = -1;
                for (int j = n - 1; j >= 0; j--)
                        if (j != drop) dm[(size_t)keep * n + j] = (dm[(size_t)keep * n + j] + dm[(size_t)drop * n + j]) * 0.5f + 0.001f;
                dm[(size_t)keep * n + keep] = 0.0f;
                for (int j = n - 1; j >= 0; j--) dm[(size_t)j * n + keep] = dm[(size_t)keep * n + j];
        }
        /* post-order over the binary tree rooted at slot[keep]: children first, then the node gets its label and task */
        int sp = 0, lab = n, nt = 0;
        for (int i = 0; i < n; i++) label[i] = i;
        stack[sp++] = slot[keep]; stack[sp++] = 0;
        while (sp) {
                const int state = stack[--sp], node = stack[--sp];
                if (left[node] < 0) continue;
                if (state == 0) {
                        stack[sp++] = node; stack[sp++] = 1;
                        stack[sp++] = right[node]; stack[sp++] = 0;      /* popped after the left subtree is done */
                        stack[sp++] = left[node]; stack[sp++] = 0;
                } else {
                        label[node] = lab++;
                        tasks_abc[3 * nt] = label[left[node]];
                        tasks_abc[3 * nt + 1] = label[right[node]];
                        tasks_abc[3 * nt + 2] = label[node];
                        nt++;
                }
        }
        free(left); free(right); free(slot); free(label); free(stack);
        return nt == n - 1 ? 0 : 1;
}
