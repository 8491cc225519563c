/* solvertypes.h -- C-visible solver selector of the BLASTed API (values match the reference's
 * include/solvertypes.h:14-26 so that a Blasted_data struct filled by an application keeps working). */
#ifndef BLASTED_SOLVERTYPES_H
#define BLASTED_SOLVERTYPES_H
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	BLASTED_JACOBI,           /* (block-)Jacobi                                  */
	BLASTED_GS,               /* chaotic forward Gauss-Seidel relaxation         */
	BLASTED_SGS,              /* asynchronous (block-)SGS                        */
	BLASTED_ILU0,             /* asynchronous (block-)ILU(0), async factor+apply */
	BLASTED_SEQILU0,          /* sequential factor, sequential apply             */
	BLASTED_SFILU0,           /* sequential factor, async apply                  */
	BLASTED_SAPILU0,          /* async factor, sequential apply                  */
	BLASTED_CSC_BGS,
	BLASTED_LEVEL_SGS,
	BLASTED_ASYNC_LEVEL_ILU0,
	BLASTED_NO_PREC,
	BLASTED_EXTERNAL
} BlastedSolverType;

/* sweep count that requests the sequential variant */
#define BLASTED_SEQUENTIAL_SYMBOL -1

#ifdef __cplusplus
}
#endif
#endif
