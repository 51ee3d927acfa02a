// Aligned rows from the residue->column tables: finalise_alignment / make_linear_sequence
// (reference lib/src/msa_op.c:546-598) with the loop over residues spread over threads.  Byte work, bound by the
// N x alnlen bytes it writes; one workgroup per sequence.
#include <hip/hip_runtime.h>
#include <stdint.h>

// Thread p owns residue p of its sequence: the gap run in front of it (columns colof[p-1]+1 .. colof[p]-1), the
// letter itself, and -- the last residue -- the trailing gaps and the terminator.  Every byte of the row is
// written exactly once.
__global__ void __launch_bounds__(256) ka_rows_kernel(const uint8_t* __restrict__ letters, const int* __restrict__ off,
                                                      const int* __restrict__ lens, const int* __restrict__ colof,
                                                      const int* __restrict__ alnlen, uint8_t gap,
                                                      uint8_t* __restrict__ rows, long long stride)
{
        const int i = blockIdx.x;
        const int len = lens[i];
        const int n = alnlen[i];
        const uint8_t* s = letters + off[i];
        const int* col = colof + off[i];
        uint8_t* row = rows + (long long)i * stride;
        for (int p = threadIdx.x; p < len; p += blockDim.x) {
                const int c = col[p];
                const int first = p ? col[p - 1] + 1 : 0;
                for (int j = first; j < c; j++) row[j] = gap;
                row[c] = s[p];
                if (p == len - 1) {
                        for (int j = c + 1; j < n; j++) row[j] = gap;
                        row[n] = 0;
                }
        }
}

extern "C" void ka_launch_rows(const uint8_t* letters, const int* off, const int* lens, const int* colof, const int* alnlen,
                               int numseq, uint8_t gap, uint8_t* rows, long long stride, hipStream_t stream)
{
        hipLaunchKernelGGL(ka_rows_kernel, dim3(numseq), dim3(256), 0, stream, letters, off, lens, colof, alnlen, gap, rows, stride);
}
