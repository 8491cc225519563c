/* blasted_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See blasted_oracle.h for scope, the
 * parity statement and the reference citations of every entry point.
 *
 * Build: make -C oracle   (gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC)
 */
#include "blasted_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXBS 16
#define AINL static inline __attribute__((always_inline))

int orc_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

void orc_set_num_threads(int n)
{
	if (n > 0)
		omp_set_num_threads(n);
}

/* ---------------------------------------------------------------- dense block helpers */

/* entry (r,c) of a block */
#define BIDX(r, c, bs, rm) ((rm) ? ((r) * (bs) + (c)) : ((c) * (bs) + (r)))

/* acc += A*x */
AINL void blk_matvec_acc(const int bs, const int rm, const double *A, const double *x, double *acc)
{
	for (int r = 0; r < bs; r++) {
		double s = 0;
		for (int c = 0; c < bs; c++)
			s += A[BIDX(r, c, bs, rm)] * x[c];
		acc[r] += s;
	}
}

/* out = A*x */
AINL void blk_matvec(const int bs, const int rm, const double *A, const double *x, double *out)
{
	for (int r = 0; r < bs; r++) {
		double s = 0;
		for (int c = 0; c < bs; c++)
			s += A[BIDX(r, c, bs, rm)] * x[c];
		out[r] = s;
	}
}

/* S -= L*U */
AINL void blk_gemm_sub(const int bs, const int rm, const double *L, const double *U, double *S)
{
	for (int r = 0; r < bs; r++)
		for (int c = 0; c < bs; c++) {
			double s = 0;
			for (int k = 0; k < bs; k++)
				s += L[BIDX(r, k, bs, rm)] * U[BIDX(k, c, bs, rm)];
			S[BIDX(r, c, bs, rm)] -= s;
		}
}

/* C = A*B (C must not alias A or B) */
AINL void blk_gemm(const int bs, const int rm, const double *A, const double *B, double *C)
{
	for (int r = 0; r < bs; r++)
		for (int c = 0; c < bs; c++) {
			double s = 0;
			for (int k = 0; k < bs; k++)
				s += A[BIDX(r, k, bs, rm)] * B[BIDX(k, c, bs, rm)];
			C[BIDX(r, c, bs, rm)] = s;
		}
}

/* kernels_ilu0_factorize.hpp:61-69 */
AINL void blk_scale(const int bs, const int rm, const double *scale, const int brow, const int bcol,
                    double *blk)
{
	for (int j = 0; j < bs; j++)
		for (int i = 0; i < bs; i++)
			blk[BIDX(i, j, bs, rm)] *= scale[brow * bs + i] * scale[bcol * bs + j];
}

/* n<=4: adjugate / determinant, the closed form Eigen's fixed-size inverse() also uses for n<=4 */
static int cofactor_inverse(int bs, int rm, const double *a, double *ainv)
{
	double A[4][4], C[4][4];
	for (int r = 0; r < bs; r++)
		for (int c = 0; c < bs; c++)
			A[r][c] = a[BIDX(r, c, bs, rm)];
	double det;
	if (bs == 1) {
		det = A[0][0];
		C[0][0] = 1.0;
	} else if (bs == 2) {
		det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
		C[0][0] = A[1][1]; C[0][1] = -A[1][0]; C[1][0] = -A[0][1]; C[1][1] = A[0][0];
	} else {
		/* cofactor C[r][c] = (-1)^(r+c) * minor(r,c) */
		for (int r = 0; r < bs; r++)
			for (int c = 0; c < bs; c++) {
				double M[3][3];
				int mr = 0;
				for (int i = 0; i < bs; i++) {
					if (i == r) continue;
					int mc = 0;
					for (int j = 0; j < bs; j++) {
						if (j == c) continue;
						M[mr][mc++] = A[i][j];
					}
					mr++;
				}
				double minor;
				if (bs == 3)
					minor = M[0][0] * M[1][1] - M[0][1] * M[1][0];
				else
					minor = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1])
					      - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0])
					      + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
				C[r][c] = ((r + c) & 1) ? -minor : minor;
			}
		det = 0;
		for (int c = 0; c < bs; c++)
			det += A[0][c] * C[0][c];
	}
	const double invdet = 1.0 / det;
	for (int r = 0; r < bs; r++)
		for (int c = 0; c < bs; c++)
			ainv[BIDX(r, c, bs, rm)] = C[c][r] * invdet;
	return det == 0.0;
}

int orc_block_inverse(int bs, int rowmajor, const double *a, double *ainv)
{
	if (bs < 1 || bs > ORC_MAXBS)
		return 1;
	if (bs <= 4)
		return cofactor_inverse(bs, rowmajor, a, ainv);
	double w[ORC_MAXBS][2 * ORC_MAXBS];
	for (int r = 0; r < bs; r++) {
		for (int c = 0; c < bs; c++) {
			w[r][c] = a[BIDX(r, c, bs, rowmajor)];
			w[r][bs + c] = (r == c) ? 1.0 : 0.0;
		}
	}
	int singular = 0;
	for (int k = 0; k < bs; k++) {
		int p = k;
		double best = fabs(w[k][k]);
		for (int r = k + 1; r < bs; r++)
			if (fabs(w[r][k]) > best) {
				best = fabs(w[r][k]);
				p = r;
			}
		if (best == 0.0)
			singular = 1;
		if (p != k)
			for (int c = 0; c < 2 * bs; c++) {
				const double t = w[k][c];
				w[k][c] = w[p][c];
				w[p][c] = t;
			}
		const double piv = 1.0 / w[k][k];
		for (int c = 0; c < 2 * bs; c++)
			w[k][c] *= piv;
		for (int r = 0; r < bs; r++) {
			if (r == k)
				continue;
			const double f = w[r][k];
			if (f != 0.0)
				for (int c = 0; c < 2 * bs; c++)
					w[r][c] -= f * w[k][c];
		}
	}
	for (int r = 0; r < bs; r++)
		for (int c = 0; c < bs; c++)
			ainv[BIDX(r, c, bs, rowmajor)] = w[r][bs + c];
	return singular;
}

/* ---------------------------------------------------------------- ILU(0) position lists */

/* helper_algorithms.hpp:38-49 */
static inline int inner_search(const int *aind, int start, int end, int tofind)
{
	for (int j = start; j < end; j++)
		if (aind[j] == tofind)
			return j;
	return -1;
}

/* One routine does both passes of ilu_pattern.cpp:47-86 (count) and :102-157 (fill). */
static long ilu_positions_pass(const orc_bsr *m, int *posptr, const int *posptr_in, int *lowerp,
                               int *upperp)
{
	long total = 0;
	for (int irow = 0; irow < m->nbrows; irow++) {
		for (int j = m->browptr[irow]; j < m->browptr[irow + 1]; j++) {
			const int colj = m->bcolind[j];
			/* l_ij: k runs over columns < j ; u_ij: k runs over columns < i */
			const int klimit = (irow > colj) ? colj : irow;
			int cnt = 0;
			for (int k = m->browptr[irow]; k < m->browptr[irow + 1] && m->bcolind[k] < klimit; k++) {
				const int krow = m->bcolind[k];
				const int ipos = inner_search(m->bcolind, m->diagind[krow], m->browptr[krow + 1], colj);
				if (ipos > -1) {
					if (lowerp) {
						lowerp[posptr_in[j] + cnt] = k;
						upperp[posptr_in[j] + cnt] = ipos;
					}
					cnt++;
				}
			}
			if (posptr)
				posptr[j + 1] = cnt;
			total += cnt;
		}
	}
	return total;
}

long orc_ilu_positions_count(const orc_bsr *m, int *posptr)
{
	posptr[0] = 0;
	const long total = ilu_positions_pass(m, posptr, NULL, NULL, NULL);
	/* helper_algorithms.cpp inclusive_scan */
	const int n = m->browptr[m->nbrows];
	for (int j = 0; j < n; j++)
		posptr[j + 1] += posptr[j];
	return total;
}

void orc_ilu_positions_fill(const orc_bsr *m, const int *posptr, int *lowerp, int *upperp)
{
	ilu_positions_pass(m, NULL, posptr, lowerp, upperp);
}

void orc_scaling_vector(const orc_bsr *m, double *scale)
{
	const int bs = m->bs, bs2 = bs * bs;
#pragma omp parallel for
	for (int i = 0; i < m->nbrows; i++)
		for (int j = 0; j < bs; j++)
			scale[(long)i * bs + j] = 1.0 / sqrt(m->vals[(long)m->diagind[i] * bs2 + j * bs + j]);
}

/* ---------------------------------------------------------------- ILU(0) factorization */

/* kernels_ilu0_factorize.hpp:19-53 : one row of the scalar fixed-point map, reads `in`, writes `out` */
AINL void scalar_factor_row(const orc_bsr *m, const int *posptr, const int *lowerp, const int *upperp,
                            const double *scale, const int irow, const double *in, double *out)
{
	for (int j = m->browptr[irow]; j < m->browptr[irow + 1]; j++) {
		double sum = m->vals[j];
		if (scale) {
			sum *= scale[irow];
			sum *= scale[m->bcolind[j]];
		}
		for (int k = posptr[j]; k < posptr[j + 1]; k++)
			sum -= in[lowerp[k]] * in[upperp[k]];
		if (irow > m->bcolind[j])
			sum = sum / in[m->diagind[m->bcolind[j]]];
		out[j] = sum;
	}
}

/* kernels_ilu0_factorize.hpp:71-98 */
AINL void block_factor_row(const orc_bsr *m, const int bs, const int rm, const int *posptr,
                           const int *lowerp, const int *upperp, const double *scale, const int irow,
                           const double *in, double *out)
{
	const int bs2 = bs * bs;
	double sum[ORC_MAXBS * ORC_MAXBS], inv[ORC_MAXBS * ORC_MAXBS], res[ORC_MAXBS * ORC_MAXBS];
	for (int jpos = m->browptr[irow]; jpos < m->browptr[irow + 1]; jpos++) {
		const int column = m->bcolind[jpos];
		for (int e = 0; e < bs2; e++)
			sum[e] = m->vals[(long)jpos * bs2 + e];
		if (scale)
			blk_scale(bs, rm, scale, irow, column, sum);
		for (int k = posptr[jpos]; k < posptr[jpos + 1]; k++)
			blk_gemm_sub(bs, rm, in + (long)lowerp[k] * bs2, in + (long)upperp[k] * bs2, sum);
		if (irow > column) {
			orc_block_inverse(bs, rm, in + (long)m->diagind[column] * bs2, inv);
			blk_gemm(bs, rm, sum, inv, res);
			for (int e = 0; e < bs2; e++)
				out[(long)jpos * bs2 + e] = res[e];
		} else {
			for (int e = 0; e < bs2; e++)
				out[(long)jpos * bs2 + e] = sum[e];
		}
	}
}

static void factor_row_dispatch(const orc_bsr *m, const int *posptr, const int *lowerp,
                                const int *upperp, const double *scale, const int irow,
                                const double *in, double *out)
{
	switch (m->bs) {
	case 1: scalar_factor_row(m, posptr, lowerp, upperp, scale, irow, in, out); break;
	case 4: block_factor_row(m, 4, m->rowmajor, posptr, lowerp, upperp, scale, irow, in, out); break;
	case 5: block_factor_row(m, 5, m->rowmajor, posptr, lowerp, upperp, scale, irow, in, out); break;
	case 8: block_factor_row(m, 8, m->rowmajor, posptr, lowerp, upperp, scale, irow, in, out); break;
	default: block_factor_row(m, m->bs, m->rowmajor, posptr, lowerp, upperp, scale, irow, in, out);
	}
}

/* async_blockilu_factor.cpp:206-254 and async_ilu_factor.cpp:109-151.
 * Deviation, scalar + scaling: the reference indexes scale[] with diagind[col] (a storage position,
 * async_ilu_factor.cpp:120-122), which reads out of bounds; the intended scale[col] is used here. */
static void fact_init_sgs(const orc_bsr *m, const double *scale, double *iluvals)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
	if (bs == 1) {
		for (int i = 0; i < m->nbrows; i++) {
			for (int j = m->browptr[i]; j < m->browptr[i + 1]; j++)
				iluvals[j] = scale ? scale[i] * m->vals[j] * scale[m->bcolind[j]] : m->vals[j];
			for (int j = m->browptr[i]; j < m->diagind[i]; j++) {
				const int col = m->bcolind[j];
				if (scale)
					iluvals[j] *= 1.0 / (m->vals[m->diagind[col]] * scale[col] * scale[col]);
				else
					iluvals[j] *= 1.0 / m->vals[m->diagind[col]];
			}
		}
		return;
	}
	double *dblks = (double *)malloc(sizeof(double) * (size_t)m->nbrows * bs2);
	double tmp[ORC_MAXBS * ORC_MAXBS];
	for (int i = 0; i < m->nbrows; i++) {
		for (int e = 0; e < bs2; e++)
			tmp[e] = m->vals[(long)m->diagind[i] * bs2 + e];
		if (scale)
			blk_scale(bs, rm, scale, i, i, tmp);
		orc_block_inverse(bs, rm, tmp, dblks + (long)i * bs2);
		for (int j = m->browptr[i]; j < m->browptr[i + 1]; j++) {
			for (int e = 0; e < bs2; e++)
				iluvals[(long)j * bs2 + e] = m->vals[(long)j * bs2 + e];
			if (scale)
				blk_scale(bs, rm, scale, i, m->bcolind[j], iluvals + (long)j * bs2);
		}
	}
	for (int i = 0; i < m->nbrows; i++)
		for (int j = m->browptr[i]; j < m->diagind[i]; j++) {
			blk_gemm(bs, rm, iluvals + (long)j * bs2, dblks + (long)m->bcolind[j] * bs2, tmp);
			for (int e = 0; e < bs2; e++)
				iluvals[(long)j * bs2 + e] = tmp[e];
		}
	free(dblks);
}

static void fact_init_original(const orc_bsr *m, const double *scale, double *iluvals)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
	const long nv = (long)m->browptr[m->nbrows] * bs2;
	if (!scale) {
		for (long i = 0; i < nv; i++)
			iluvals[i] = m->vals[i];
		return;
	}
	for (int i = 0; i < m->nbrows; i++)
		for (int j = m->browptr[i]; j < m->browptr[i + 1]; j++) {
			if (bs == 1)
				iluvals[j] = scale[i] * m->vals[j] * scale[m->bcolind[j]];
			else {
				for (int e = 0; e < bs2; e++)
					iluvals[(long)j * bs2 + e] = m->vals[(long)j * bs2 + e];
				blk_scale(bs, rm, scale, i, m->bcolind[j], iluvals + (long)j * bs2);
			}
		}
}

double orc_ilu0_nonlinear_res(const orc_bsr *m, const int *posptr, const int *lowerp,
                              const int *upperp, const double *scale, const double *ilu)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
	double resnorm = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : resnorm)
	for (int irow = 0; irow < m->nbrows; irow++) {
		double sum[ORC_MAXBS * ORC_MAXBS];
		for (int jj = m->browptr[irow]; jj < m->browptr[irow + 1]; jj++) {
			const int col = m->bcolind[jj];
			for (int e = 0; e < bs2; e++)
				sum[e] = m->vals[(long)jj * bs2 + e];
			if (scale) {
				if (bs == 1) {
					sum[0] *= scale[irow];
					sum[0] *= scale[col];
				} else
					blk_scale(bs, rm, scale, irow, col, sum);
			}
			for (int k = posptr[jj]; k < posptr[jj + 1]; k++)
				blk_gemm_sub(bs, rm, ilu + (long)lowerp[k] * bs2, ilu + (long)upperp[k] * bs2, sum);
			if (irow > col)
				blk_gemm_sub(bs, rm, ilu + (long)jj * bs2, ilu + (long)m->diagind[col] * bs2, sum);
			else
				for (int e = 0; e < bs2; e++)
					sum[e] -= ilu[(long)jj * bs2 + e];
			double blockres = 0;
			for (int e = 0; e < bs2; e++)
				blockres += fabs(sum[e]);
			resnorm += blockres;
		}
	}
	return resnorm;
}

void orc_diag_dominance(const orc_bsr *m, double *out4)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
	double uddavg = 0, uddmin = 1e30, lddavg = 0, lddmin = 1e30;
	for (int irow = 0; irow < m->nbrows; irow++) {
		double rowddu[ORC_MAXBS], rowddl[ORC_MAXBS];
		for (int i = 0; i < bs; i++)
			rowddl[i] = rowddu[i] = 0;
		const int diagp = m->diagind[irow];
		const double *dblk = m->vals + (long)diagp * bs2;
		for (int i = 0; i < bs; i++)
			for (int j = 0; j < bs; j++)
				if (i != j)
					rowddu[i] += fabs(dblk[BIDX(i, j, bs, rm)]);
		for (int jj = diagp + 1; jj < m->browptr[irow + 1]; jj++)
			for (int i = 0; i < bs; i++)
				for (int j = 0; j < bs; j++)
					rowddu[i] += fabs(m->vals[(long)jj * bs2 + BIDX(i, j, bs, rm)]);
		for (int jj = m->browptr[irow]; jj < diagp; jj++)
			for (int i = 0; i < bs; i++)
				for (int j = 0; j < bs; j++)
					rowddl[i] += fabs(m->vals[(long)jj * bs2 + BIDX(i, j, bs, rm)]);
		for (int i = 0; i < bs; i++) {
			rowddl[i] = 1.0 - rowddl[i];
			rowddu[i] = 1.0 - rowddu[i] / fabs(dblk[BIDX(i, i, bs, rm)]);
			if (uddmin > rowddu[i])
				uddmin = rowddu[i];
			if (lddmin > rowddl[i])
				lddmin = rowddl[i];
			lddavg += rowddl[i];
			uddavg += rowddu[i];
		}
	}
	out4[0] = lddavg / ((double)m->nbrows * bs);
	out4[1] = lddmin;
	out4[2] = uddavg / ((double)m->nbrows * bs);
	out4[3] = uddmin;
}

int orc_ilu0_factorize(const orc_bsr *m, const int *posptr, const int *lowerp, const int *upperp,
                       int nbuildsweeps, int chunk, int mode, int init_type, double *iluvals,
                       double *scale, double *precinfo)
{
	const int bs = m->bs, bs2 = bs * bs;
	const long nv = (long)m->browptr[m->nbrows] * bs2;
	if (bs < 1 || bs > ORC_MAXBS || chunk < 1)
		return -1;
	/* ORC_KEEP_DIAG (tests): leave the diagonal blocks as the sweeps iterate on them -- the form the reference's own
	 * fixed-point tests compare (tests/solverops/async_ilu_convergence.cpp drives the kernels and never inverts) */
	const int keep_diag = init_type & ORC_KEEP_DIAG;
	init_type &= ~ORC_KEEP_DIAG;

	if (scale)
		orc_scaling_vector(m, scale);

	switch (init_type) {
	case ORC_INIT_F_ZERO:
		for (long i = 0; i < nv; i++)
			iluvals[i] = 0;
		if (bs > 1)
			break;
		/* scalar: missing `break` in the reference, async_ilu_factor.cpp:48-54 */
		/* fall through */
	case ORC_INIT_F_ORIGINAL: fact_init_original(m, scale, iluvals); break;
	case ORC_INIT_F_SGS: fact_init_sgs(m, scale, iluvals); break;
	default:;
	}

	if (precinfo) {
		for (int i = 0; i < 6; i++)
			precinfo[i] = 0;
		precinfo[1] = orc_ilu0_nonlinear_res(m, posptr, lowerp, upperp, scale, iluvals);
	}

	if (mode == ORC_GS_SERIAL) {
		for (int isweep = 0; isweep < nbuildsweeps; isweep++)
			for (int irow = 0; irow < m->nbrows; irow++)
				factor_row_dispatch(m, posptr, lowerp, upperp, scale, irow, iluvals, iluvals);
	} else if (mode == ORC_JACOBI_SYNC) {
		double *other = (double *)malloc(sizeof(double) * (size_t)nv);
		double *in = iluvals, *out = other;
		for (int isweep = 0; isweep < nbuildsweeps; isweep++) {
#pragma omp parallel for schedule(static)
			for (int irow = 0; irow < m->nbrows; irow++)
				factor_row_dispatch(m, posptr, lowerp, upperp, scale, irow, in, out);
			double *t = in;
			in = out;
			out = t;
		}
		if (in != iluvals)
			memcpy(iluvals, in, sizeof(double) * (size_t)nv);
		free(other);
	} else if (mode == ORC_ASYNC_OMP) {
		/* async_blockilu_factor.cpp:196-203 */
#pragma omp parallel default(shared)
		for (int isweep = 0; isweep < nbuildsweeps; isweep++) {
#pragma omp for schedule(dynamic, chunk) nowait
			for (int irow = 0; irow < m->nbrows; irow++)
				factor_row_dispatch(m, posptr, lowerp, upperp, scale, irow, iluvals, iluvals);
		}
	} else
		return -1;

	if (precinfo) {
		precinfo[0] = orc_ilu0_nonlinear_res(m, posptr, lowerp, upperp, scale, iluvals);
		orc_bsr f = *m;
		f.vals = iluvals;
		double dd[4];
		orc_diag_dominance(&f, dd);
		precinfo[5] = dd[0]; /* lower avg */
		precinfo[4] = dd[1]; /* lower min */
		precinfo[3] = dd[2]; /* upper avg */
		precinfo[2] = dd[3]; /* upper min */
	}

	if (bs > 1 && !keep_diag) {
		/* async_blockilu_factor.cpp:143-146 */
#pragma omp parallel for
		for (int irow = 0; irow < m->nbrows; irow++) {
			double inv[ORC_MAXBS * ORC_MAXBS];
			double *d = iluvals + (long)m->diagind[irow] * bs2;
			orc_block_inverse(bs, m->rowmajor, d, inv);
			for (int e = 0; e < bs2; e++)
				d[e] = inv[e];
		}
	}
	return 0;
}

/* ---------------------------------------------------------------- triangular sweeps */

/* kernels_ilu_apply.hpp:54-67 (bs>1) and :15-27 (bs==1): xout_i = rhs_i - sum_{j<diag} L_ij xin_j */
AINL void lower_row(const orc_bsr *m, const int bs, const int rm, const double *vals, const int i,
                    const double *rhs, const double *xin, double *xout)
{
	const int bs2 = bs * bs;
	double inter[ORC_MAXBS];
	for (int r = 0; r < bs; r++)
		inter[r] = 0;
	for (int jj = m->browptr[i]; jj < m->diagind[i]; jj++)
		blk_matvec_acc(bs, rm, vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	for (int r = 0; r < bs; r++)
		xout[(long)i * bs + r] = rhs[(long)i * bs + r] - inter[r];
}

/* kernels_ilu_apply.hpp:79-94 (diag block pre-inverted) and :30-42 with 1/ilu[diag]
 * (solverops_ilu0.cpp:311-312): xout_i = Dinv_i (rhs_i - sum_{j>diag} U_ij xin_j) */
AINL void upper_row(const orc_bsr *m, const int bs, const int rm, const double *vals, const int i,
                    const double *rhs, const double *xin, double *xout)
{
	const int bs2 = bs * bs;
	double inter[ORC_MAXBS], t[ORC_MAXBS], o[ORC_MAXBS];
	for (int r = 0; r < bs; r++)
		inter[r] = 0;
	for (int jj = m->diagind[i] + 1; jj < m->browptr[i + 1]; jj++)
		blk_matvec_acc(bs, rm, vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	if (bs == 1) {
		xout[i] = (1.0 / vals[m->diagind[i]]) * (rhs[i] - inter[0]);
		return;
	}
	for (int r = 0; r < bs; r++)
		t[r] = rhs[(long)i * bs + r] - inter[r];
	blk_matvec(bs, rm, vals + (long)m->diagind[i] * bs2, t, o);
	for (int r = 0; r < bs; r++)
		xout[(long)i * bs + r] = o[r];
}

#define DISPATCH_BS(FN, ...)                                      \
	switch (m->bs) {                                              \
	case 1: FN(m, 1, 0, __VA_ARGS__); break;                      \
	case 4: FN(m, 4, m->rowmajor, __VA_ARGS__); break;            \
	case 5: FN(m, 5, m->rowmajor, __VA_ARGS__); break;            \
	case 8: FN(m, 8, m->rowmajor, __VA_ARGS__); break;            \
	default: FN(m, m->bs, m->rowmajor, __VA_ARGS__);              \
	}

static void lower_row_d(const orc_bsr *m, const double *vals, int i, const double *rhs,
                        const double *xin, double *xout)
{
	DISPATCH_BS(lower_row, vals, i, rhs, xin, xout)
}
static void upper_row_d(const orc_bsr *m, const double *vals, int i, const double *rhs,
                        const double *xin, double *xout)
{
	DISPATCH_BS(upper_row, vals, i, rhs, xin, xout)
}

typedef void (*rowfn)(const orc_bsr *, const double *, int, const double *, const double *, double *);

/* `nsweeps` sweeps of x_i <- f(rhs_i, x) over all rows, ascending or descending, in `mode`.
 * vals2 is the second value array some row functions need (dblocks); passed through `vals`. */
static void run_sweeps(const orc_bsr *m, rowfn fn, const double *vals, const double *rhs, double *x,
                       int nsweeps, int chunk, int mode, int descending)
{
	const int nb = m->nbrows;
	const long n = (long)nb * m->bs;
	if (mode == ORC_GS_SERIAL) {
		for (int s = 0; s < nsweeps; s++) {
			if (!descending)
				for (int i = 0; i < nb; i++)
					fn(m, vals, i, rhs, x, x);
			else
				for (int i = nb - 1; i >= 0; i--)
					fn(m, vals, i, rhs, x, x);
		}
	} else if (mode == ORC_JACOBI_SYNC) {
		double *other = (double *)malloc(sizeof(double) * (size_t)n);
		double *in = x, *out = other;
		for (int s = 0; s < nsweeps; s++) {
#pragma omp parallel for schedule(static)
			for (int i = 0; i < nb; i++)
				fn(m, vals, i, rhs, in, out);
			double *t = in;
			in = out;
			out = t;
		}
		if (in != x)
			memcpy(x, in, sizeof(double) * (size_t)n);
		free(other);
	} else {
		/* solverops_ilu0.cpp:99-108,132-141 */
#pragma omp parallel default(shared)
		for (int s = 0; s < nsweeps; s++) {
			if (!descending) {
#pragma omp for schedule(dynamic, chunk) nowait
				for (int i = 0; i < nb; i++)
					fn(m, vals, i, rhs, x, x);
			} else {
#pragma omp for schedule(dynamic, chunk) nowait
				for (int i = nb - 1; i >= 0; i--)
					fn(m, vals, i, rhs, x, x);
			}
		}
	}
}

int orc_ilu0_apply(const orc_bsr *m, const double *iluvals, const double *scale, double *ytemp,
                   int napplysweeps, int chunk, int mode, int init_type, const double *r, double *z)
{
	const long n = (long)m->nbrows * m->bs;
	if (init_type != ORC_INIT_A_ZERO && init_type != ORC_INIT_A_JACOBI)
		return -1; /* solverops_ilu0.cpp:125-126 throws */
	if (chunk < 1)
		return -1;

	/* z := S r is the right-hand side of the L solve */
	for (long i = 0; i < n; i++)
		z[i] = scale ? scale[i] * r[i] : r[i];
	for (long i = 0; i < n; i++)
		ytemp[i] = 0;

	run_sweeps(m, lower_row_d, iluvals, z, ytemp, napplysweeps, chunk, mode, 0);

	if (init_type == ORC_INIT_A_JACOBI)
		for (long i = 0; i < n; i++)
			z[i] = ytemp[i];
	else
		for (long i = 0; i < n; i++)
			z[i] = 0;

	run_sweeps(m, upper_row_d, iluvals, ytemp, z, napplysweeps, chunk, mode, 1);

	if (scale)
		for (long i = 0; i < n; i++)
			z[i] = z[i] * scale[i];
	return 0;
}

/* ---------------------------------------------------------------- Jacobi / SGS / relaxation */

int orc_jacobi_compute(const orc_bsr *m, double *dblocks)
{
	const int bs = m->bs, bs2 = bs * bs;
	int bad = 0;
#pragma omp parallel for reduction(| : bad)
	for (int i = 0; i < m->nbrows; i++) {
		if (bs == 1)
			dblocks[i] = 1.0 / m->vals[m->diagind[i]];
		else
			bad |= orc_block_inverse(bs, m->rowmajor, m->vals + (long)m->diagind[i] * bs2,
			                         dblocks + (long)i * bs2);
	}
	return bad;
}

void orc_jacobi_apply(const orc_bsr *m, const double *dblocks, const double *r, double *z)
{
	const int bs = m->bs, bs2 = bs * bs;
	for (int i = 0; i < m->nbrows; i++)
		blk_matvec(bs, m->rowmajor, dblocks + (long)i * bs2, r + (long)i * bs, z + (long)i * bs);
}

/* A bundle so that the SGS row functions fit the rowfn signature: vals points at this. */
typedef struct {
	const double *vals;
	const double *dblocks;
} sgs_vals;

/* kernels_sgs.hpp:47-60 / :17-29 : x_i = Dinv_i (rhs_i - sum_{L} A_ij x_j) */
AINL void fgs_row(const orc_bsr *m, const int bs, const int rm, const double *pv, const int i,
                  const double *rhs, const double *xin, double *xout)
{
	const sgs_vals *sv = (const sgs_vals *)pv;
	const int bs2 = bs * bs;
	double inter[ORC_MAXBS], t[ORC_MAXBS], o[ORC_MAXBS];
	for (int r = 0; r < bs; r++)
		inter[r] = 0;
	for (int jj = m->browptr[i]; jj < m->diagind[i]; jj++)
		blk_matvec_acc(bs, rm, sv->vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	for (int r = 0; r < bs; r++)
		t[r] = rhs[(long)i * bs + r] - inter[r];
	blk_matvec(bs, rm, sv->dblocks + (long)i * bs2, t, o);
	for (int r = 0; r < bs; r++)
		xout[(long)i * bs + r] = o[r];
}

/* kernels_sgs.hpp:62-76 / :31-44 : x_i = rhs_i - Dinv_i sum_{U} A_ij x_j */
AINL void bgs_row(const orc_bsr *m, const int bs, const int rm, const double *pv, const int i,
                  const double *rhs, const double *xin, double *xout)
{
	const sgs_vals *sv = (const sgs_vals *)pv;
	const int bs2 = bs * bs;
	double inter[ORC_MAXBS], o[ORC_MAXBS];
	for (int r = 0; r < bs; r++)
		inter[r] = 0;
	for (int jj = m->diagind[i] + 1; jj < m->browptr[i + 1]; jj++)
		blk_matvec_acc(bs, rm, sv->vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	blk_matvec(bs, rm, sv->dblocks + (long)i * bs2, inter, o);
	for (int r = 0; r < bs; r++)
		xout[(long)i * bs + r] = rhs[(long)i * bs + r] - o[r];
}

/* kernels_relaxation.hpp:17-54 : x_i = Dinv_i (rhs_i - sum_{j != i} A_ij x_j) */
AINL void relax_row(const orc_bsr *m, const int bs, const int rm, const double *pv, const int i,
                    const double *rhs, const double *xin, double *xout)
{
	const sgs_vals *sv = (const sgs_vals *)pv;
	const int bs2 = bs * bs;
	double inter[ORC_MAXBS], t[ORC_MAXBS], o[ORC_MAXBS];
	for (int r = 0; r < bs; r++)
		inter[r] = 0;
	for (int jj = m->browptr[i]; jj < m->diagind[i]; jj++)
		blk_matvec_acc(bs, rm, sv->vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	for (int jj = m->diagind[i] + 1; jj < m->browptr[i + 1]; jj++)
		blk_matvec_acc(bs, rm, sv->vals + (long)jj * bs2, xin + (long)m->bcolind[jj] * bs, inter);
	for (int r = 0; r < bs; r++)
		t[r] = rhs[(long)i * bs + r] - inter[r];
	blk_matvec(bs, rm, sv->dblocks + (long)i * bs2, t, o);
	for (int r = 0; r < bs; r++)
		xout[(long)i * bs + r] = o[r];
}

static void fgs_row_d(const orc_bsr *m, const double *pv, int i, const double *rhs, const double *xin,
                      double *xout)
{
	DISPATCH_BS(fgs_row, pv, i, rhs, xin, xout)
}
static void bgs_row_d(const orc_bsr *m, const double *pv, int i, const double *rhs, const double *xin,
                      double *xout)
{
	DISPATCH_BS(bgs_row, pv, i, rhs, xin, xout)
}
static void relax_row_d(const orc_bsr *m, const double *pv, int i, const double *rhs,
                        const double *xin, double *xout)
{
	DISPATCH_BS(relax_row, pv, i, rhs, xin, xout)
}

void orc_sgs_apply(const orc_bsr *m, const double *dblocks, double *ytemp, int napplysweeps,
                   int chunk, int mode, int init_type, const double *r, double *z)
{
	const long n = (long)m->nbrows * m->bs;
	sgs_vals sv = {m->vals, dblocks};
	if (init_type == ORC_INIT_A_JACOBI || init_type == ORC_INIT_A_ZERO)
		for (long i = 0; i < n; i++)
			ytemp[i] = 0;

	/* forward sweeps: serial in the reference whatever the thread count (solverops_sgs.cpp:62-66) */
	run_sweeps(m, fgs_row_d, (const double *)&sv, r, ytemp, napplysweeps, chunk,
	           mode == ORC_ASYNC_OMP ? ORC_GS_SERIAL : mode, 0);

	if (init_type == ORC_INIT_A_JACOBI)
		for (long i = 0; i < n; i++)
			z[i] = ytemp[i];
	else if (init_type == ORC_INIT_A_ZERO)
		for (long i = 0; i < n; i++)
			z[i] = 0;

	run_sweeps(m, bgs_row_d, (const double *)&sv, ytemp, z, napplysweeps, chunk, mode, 1);
}

void orc_sgs_relax(const orc_bsr *m, const double *dblocks, int maxits, int chunk, int mode,
                   const double *b, double *x)
{
	sgs_vals sv = {m->vals, dblocks};
	if (mode == ORC_ASYNC_OMP) {
		const int nb = m->nbrows;
		/* solverops_sgs.cpp:96-115 : one parallel region, both passes nowait */
#pragma omp parallel default(shared)
		for (int step = 0; step < maxits; step++) {
#pragma omp for schedule(dynamic, chunk) nowait
			for (int i = 0; i < nb; i++)
				relax_row_d(m, (const double *)&sv, i, b, x, x);
#pragma omp for schedule(dynamic, chunk) nowait
			for (int i = nb - 1; i >= 0; i--)
				relax_row_d(m, (const double *)&sv, i, b, x, x);
		}
		return;
	}
	for (int step = 0; step < maxits; step++) {
		run_sweeps(m, relax_row_d, (const double *)&sv, b, x, 1, chunk, mode, 0);
		run_sweeps(m, relax_row_d, (const double *)&sv, b, x, 1, chunk, mode, 1);
	}
}

/* src/relaxation_chaotic.cpp:21-70,92-125 : forward passes only, x in/out */
void orc_gs_relax(const orc_bsr *m, const double *dblocks, int nsweeps, int chunk, int mode,
                  const double *b, double *x)
{
	sgs_vals sv = {m->vals, dblocks};
	run_sweeps(m, relax_row_d, (const double *)&sv, b, x, nsweeps, chunk, mode, 0);
}

/* BJacobiSRPreconditioner::apply_relax, src/solverops_jacobi.cpp:66-119; returns the steps taken */
int orc_jacobi_relax(const orc_bsr *m, const double *dblocks, int maxits, int ctol, double rtol,
                     double atol, double dtol, const double *b, double *x)
{
	const long n = (long)m->nbrows * m->bs;
	sgs_vals sv = {m->vals, dblocks};
	double *xtemp = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
	double refdiffnorm = 1;
	int step = 0;
	for (; step < maxits; step++) {
#pragma omp parallel for default(shared)
		for (int i = 0; i < m->nbrows; i++)
			relax_row_d(m, (const double *)&sv, i, b, x, xtemp);
		if (ctol) {
			double diffnorm = 0;
			for (long i = 0; i < n; i++) {
				const double diff = xtemp[i] - x[i];
				diffnorm += diff * diff;
				x[i] = xtemp[i];
			}
			diffnorm = sqrt(diffnorm);
			if (step == 0)
				refdiffnorm = diffnorm;
			if (diffnorm < atol || diffnorm / refdiffnorm < rtol || diffnorm / refdiffnorm > dtol) {
				step++;
				break;
			}
		} else
			memcpy(x, xtemp, sizeof(double) * (size_t)n);
	}
	free(xtemp);
	return step;
}

/* ---------------------------------------------------------------- level scheduling */

/* computeLevels, src/levelschedule.cpp:13-72.  levels[] receives the boundaries (levels[0] = 0,
 * level l = rows [levels[l], levels[l+1])), capacity nbrows+1 entries; returns the number of levels, or
 * -1 where the reference throws "Faulty dependency list!" (structurally non-symmetric pattern).
 * The reference keeps a std::list of remaining dependencies per row and erases entries; here an erased
 * entry is a flag on the stored position and front() is the first unflagged entry of the row. */
int orc_compute_levels(const orc_bsr *m, int *levels)
{
	const int nb = m->nbrows;
	char *erased = (char *)calloc((size_t)m->browptr[nb] + 1, 1);
	int *front = (int *)malloc(sizeof(int) * ((size_t)nb + 1));
	for (int i = 0; i < nb; i++)
		front[i] = m->browptr[i];
	int inode = 0, nlevels = 0, rc = 0;
	levels[0] = 0;
	while (inode < nb && rc == 0) {
		/* 1. consecutive nodes whose smallest remaining dependency is not below themselves */
		while (inode < nb) {
			int f = front[inode];
			while (f < m->browptr[inode + 1] && erased[f])
				f++;
			front[inode] = f;
			if (f < m->browptr[inode + 1] && m->bcolind[f] < inode)
				break;
			inode++;
		}
		levels[++nlevels] = inode;
		/* 2. remove the nodes of this level from their neighbours' dependency lists */
		for (int jnode = levels[nlevels - 1]; jnode < inode && rc == 0; jnode++)
			for (int jj = m->browptr[jnode]; jj < m->browptr[jnode + 1]; jj++) {
				const int nbr = m->bcolind[jj];
				if (nbr == jnode || erased[jj])
					continue; /* a node already erased from this list is no longer iterated over */
				const int pos = inner_search(m->bcolind, m->browptr[nbr], m->browptr[nbr + 1], jnode);
				if (pos < 0 || erased[pos]) {
					rc = -1;
					break;
				}
				erased[pos] = 1;
			}
		if (levels[nlevels] == levels[nlevels - 1] && inode < nb) {
			rc = -1; /* no progress: cannot happen for a valid pattern */
		}
	}
	free(erased);
	free(front);
	return rc ? rc : nlevels;
}

/* one level-scheduled pass: `omp parallel for` over the rows of each level, levels in ascending or
 * descending order (src/solverops_levels_ilu0.cpp:81-99, src/solverops_levels_sgs.cpp:66-88) */
static void run_level_pass(const orc_bsr *m, rowfn fn, const double *vals, const double *rhs, double *x,
                           const int *levels, int nlevels, int descending)
{
	if (!descending) {
		for (int l = 0; l < nlevels; l++) {
#pragma omp parallel for default(shared)
			for (int i = levels[l]; i < levels[l + 1]; i++)
				fn(m, vals, i, rhs, x, x);
		}
	} else {
		for (int l = nlevels; l > 0; l--) {
#pragma omp parallel for default(shared)
			for (int i = levels[l] - 1; i >= levels[l - 1]; i--)
				fn(m, vals, i, rhs, x, x);
		}
	}
}

/* Async_Level_BlockILU0::apply / Async_Level_ILU0::apply, src/solverops_levels_ilu0.cpp:58-105,146-200 */
void orc_level_ilu0_apply(const orc_bsr *m, const double *iluvals, const double *scale, double *ytemp,
                          const int *levels, int nlevels, const double *r, double *z)
{
	const long n = (long)m->nbrows * m->bs;
	for (long i = 0; i < n; i++)
		z[i] = scale ? scale[i] * r[i] : r[i];
	run_level_pass(m, lower_row_d, iluvals, z, ytemp, levels, nlevels, 0);
	run_level_pass(m, upper_row_d, iluvals, ytemp, z, levels, nlevels, 1);
	if (scale)
		for (long i = 0; i < n; i++)
			z[i] = z[i] * scale[i];
}

/* Level_BSGS::apply / Level_SGS::apply, src/solverops_levels_sgs.cpp:52-88,166-195 */
void orc_level_sgs_apply(const orc_bsr *m, const double *dblocks, double *ytemp, const int *levels,
                         int nlevels, const double *r, double *z)
{
	sgs_vals sv = {m->vals, dblocks};
	run_level_pass(m, fgs_row_d, (const double *)&sv, r, ytemp, levels, nlevels, 0);
	run_level_pass(m, bgs_row_d, (const double *)&sv, ytemp, z, levels, nlevels, 1);
}

/* Level_BSGS::apply_relax / Level_SGS::apply_relax, src/solverops_levels_sgs.cpp:90-123,197-223 */
void orc_level_sgs_relax(const orc_bsr *m, const double *dblocks, const int *levels, int nlevels,
                         int maxits, const double *b, double *x)
{
	sgs_vals sv = {m->vals, dblocks};
	for (int step = 0; step < maxits; step++) {
		run_level_pass(m, relax_row_d, (const double *)&sv, b, x, levels, nlevels, 0);
		run_level_pass(m, relax_row_d, (const double *)&sv, b, x, levels, nlevels, 1);
	}
}

/* ---------------------------------------------------------------- SpMV */

void orc_spmv(const orc_bsr *m, const double *x, double *y)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
#pragma omp parallel for
	for (int i = 0; i < m->nbrows; i++) {
		double acc[ORC_MAXBS];
		for (int r = 0; r < bs; r++)
			acc[r] = 0;
		for (int jj = m->browptr[i]; jj < m->browptr[i + 1]; jj++)
			blk_matvec_acc(bs, rm, m->vals + (long)jj * bs2, x + (long)m->bcolind[jj] * bs, acc);
		for (int r = 0; r < bs; r++)
			y[(long)i * bs + r] = acc[r];
	}
}

void orc_gemv3(const orc_bsr *m, double a, const double *x, double b, const double *y, double *z)
{
	const int bs = m->bs, rm = m->rowmajor, bs2 = bs * bs;
#pragma omp parallel for
	for (int i = 0; i < m->nbrows; i++) {
		double acc[ORC_MAXBS], t[ORC_MAXBS];
		for (int r = 0; r < bs; r++)
			acc[r] = b * y[(long)i * bs + r];
		for (int jj = m->browptr[i]; jj < m->browptr[i + 1]; jj++) {
			blk_matvec(bs, rm, m->vals + (long)jj * bs2, x + (long)m->bcolind[jj] * bs, t);
			for (int r = 0; r < bs; r++)
				acc[r] += a * t[r];
		}
		for (int r = 0; r < bs; r++)
			z[(long)i * bs + r] = acc[r];
	}
}
