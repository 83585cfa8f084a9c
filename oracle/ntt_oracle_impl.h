/*
 * ntt_oracle_impl.h -- width-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by ntt_oracle.c with
 *     ORA_T   = uint32_t / uint64_t          (the reference's Data32 / Data64)
 *     ORA_T2  = uint64_t / unsigned __int128 (the reference's "T2" wide type)
 *     ORA_(x) = ora32_##x / ora64_##x
 *     ORA_TMAX= UINT32_MAX / UINT64_MAX
 *
 * Every function is a plain-C restatement of the reference algorithm it cites
 * (paths relative to /root/reference).  Nothing here is shipped in the product
 * library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call it.
 */

/* ---- Modulus<T>::bit_generator / mu_generator --------------------------------
 * src/include/gpuntt/common/modular_arith.cuh:28-57
 * bit = (T)(log2(value) + 1) evaluated in double; mu = floor(2^(2*bit+1) / value). */
void ORA_(modulus)(ORA_T q, ORA_T* bit, ORA_T* mu)
{
    ORA_T b = (ORA_T) (log2((double) q) + 1);
    ORA_T2 m = ((ORA_T2) 1) << ((2 * b) + 1);
    m = m / q;
    *bit = b;
    *mu = (ORA_T) m;
}

/* ---- BarrettOperations<T>::add  modular_arith.cuh:71-76 */
ORA_T ORA_(add)(ORA_T a, ORA_T b, ORA_T q)
{
    ORA_T sum = a + b;
    return (sum >= q) ? (sum - q) : sum;
}

/* ---- BarrettOperations<T>::sub  modular_arith.cuh:80-86 */
ORA_T ORA_(sub)(ORA_T a, ORA_T b, ORA_T q)
{
    ORA_T dif = a + q;
    dif = dif - b;
    return (dif >= q) ? (dif - q) : dif;
}

/* ---- BarrettOperations<T>::mult  modular_arith.cuh:90-107
 * z = a*b; r = ((z >> (bit-2)) * mu) >> (bit+3); z -= r*q; one conditional subtract. */
ORA_T ORA_(mult)(ORA_T a, ORA_T b, ORA_T q, ORA_T bit, ORA_T mu)
{
    ORA_T2 mult = (ORA_T2) a * (ORA_T2) b;
    ORA_T2 r = mult >> (bit - 2);
    r = r * (ORA_T2) mu;
    r = r >> (bit + 3);
    r = r * (ORA_T2) q;
    mult = mult - r;
    ORA_T result = (ORA_T) (mult & ORA_TMAX);
    return (result >= q) ? (result - q) : result;
}

/* ---- BarrettOperations<T>::exp  modular_arith.cuh:111-129
 * left-to-right square & multiply; exponent_bit from double log2 like the reference. */
ORA_T ORA_(exp)(ORA_T base, ORA_T exponent, ORA_T q, ORA_T bit, ORA_T mu)
{
    ORA_T result = 1;
    if (exponent == 0)
        return result;
    int exponent_bit = (int) (log2((double) exponent) + 1);
    for (int i = exponent_bit - 1; i >= 0; i--)
    {
        result = ORA_(mult)(result, result, q, bit, mu);
        if ((exponent >> i) & 1)
            result = ORA_(mult)(result, base, q, bit, mu);
    }
    return result;
}

/* ---- BarrettOperations<T>::modinv  modular_arith.cuh:133-137  (a^(q-2)) */
ORA_T ORA_(modinv)(ORA_T a, ORA_T q, ORA_T bit, ORA_T mu)
{
    return ORA_(exp)(a, q - 2, q, bit, mu);
}

/* ---- NTTParameters<T> pool  src/lib/common/nttparameters.cu:84-142
 * out = {q, omega, psi}.  u32: q=469762049, w=900, psi=30, top logn 25;
 * u64: q=576460756061519873, w=229929041166717729, psi=4517306222, top logn 28. */
void ORA_(merge_pool)(int logn, ORA_T* q_out, ORA_T* omega_out, ORA_T* psi_out)
{
    ORA_T q, bit, mu, w, p;
    int top;
#if ORA_BITS == 32
    q = 469762049u;
    w = 900u;
    p = 30u;
    top = 25;
#else
    q = 576460756061519873ULL;
    w = 229929041166717729ULL;
    p = 4517306222ULL;
    top = 28;
#endif
    ORA_(modulus)(q, &bit, &mu);
    *q_out = q;
    *omega_out = ORA_(exp)(w, (ORA_T) (1 << (top - logn)), q, bit, mu);
    *psi_out = ORA_(exp)(p, (ORA_T) (1 << (top - logn)), q, bit, mu);
}

/* ---- forward/inverse_root_of_unity_table_generator  nttparameters.cu:144-168
 * natural-order powers root^0 .. root^(size-1) by repeated multiplication. */
void ORA_(power_table)(ORA_T root, uint64_t size, ORA_T q, ORA_T bit, ORA_T mu,
                       ORA_T* out)
{
    out[0] = 1;
    for (uint64_t i = 1; i < size; i++)
        out[i] = ORA_(mult)(out[i - 1], root, q, bit, mu);
}

/* ---- gpu_root_of_unity_table_generator  nttparameters.cu:175-189 and :453-466
 * new_table[i] = table[bitreverse(i, log2(size))]. */
void ORA_(bitrev_table)(const ORA_T* in, uint64_t size, ORA_T* out)
{
    int lg = (int) log2((double) size);
    for (uint64_t i = 0; i < size; i++)
        out[i] = in[ora_bitreverse((int) i, lg)];
}

/* ---- NTTCPU<T>::ntt  src/lib/ntt_merge/ntt_cpu.cu:81-128
 * poly: 0 = X_N_plus (negacyclic, psi table of N entries),
 *       1 = X_N_minus (cyclic, omega table of N/2 entries)  [enum order nttparameters.cuh:32-36]
 * table = NATURAL-order forward_root_of_unity_table. In place on data[0..n). */
void ORA_(merge_ntt)(ORA_T* data, int logn, int poly, const ORA_T* table, ORA_T q,
                     ORA_T bit, ORA_T mu)
{
    int n = 1 << logn;
    int t = n;
    int m = 1;
    while (m < n)
    {
        t = t >> 1;
        for (int i = 0; i < m; i++)
        {
            int j1 = 2 * i * t;
            int j2 = j1 + t - 1;
            int index;
            if (poly == 1)
                index = ora_bitreverse(i, logn - 1);
            else
                index = ora_bitreverse(m + i, logn);
            ORA_T S = table[index];
            for (int j = j1; j < (j2 + 1); j++)
            {
                ORA_T U = data[j];
                ORA_T V = ORA_(mult)(data[j + t], S, q, bit, mu);
                data[j] = ORA_(add)(U, V, q);
                data[j + t] = ORA_(sub)(U, V, q);
            }
        }
        m = m << 1;
    }
}

/* ---- NTTCPU<T>::intt  ntt_cpu.cu:130-185  (GS butterflies, then * n^-1) */
void ORA_(merge_intt)(ORA_T* data, int logn, int poly, const ORA_T* table, ORA_T q,
                      ORA_T bit, ORA_T mu)
{
    int n = 1 << logn;
    int t = 1;
    int m = n;
    while (m > 1)
    {
        int j1 = 0;
        int h = m >> 1;
        for (int i = 0; i < h; i++)
        {
            int j2 = j1 + t - 1;
            int index;
            if (poly == 1)
                index = ora_bitreverse(i, logn - 1);
            else
                index = ora_bitreverse(h + i, logn);
            ORA_T S = table[index];
            for (int j = j1; j < (j2 + 1); j++)
            {
                ORA_T U = data[j];
                ORA_T V = data[j + t];
                data[j] = ORA_(add)(U, V, q);
                data[j + t] = ORA_(sub)(U, V, q);
                data[j + t] = ORA_(mult)(data[j + t], S, q, bit, mu);
            }
            j1 = j1 + (t << 1);
        }
        t = t << 1;
        m = m >> 1;
    }
    ORA_T n_inv = ORA_(modinv)((ORA_T) n, q, bit, mu);
    for (int i = 0; i < n; i++)
        data[i] = ORA_(mult)(data[i], n_inv, q, bit, mu);
}

/* ---- NTTCPU<T>::mult  ntt_cpu.cu:67-79  (pointwise product) */
void ORA_(pointwise)(const ORA_T* a, const ORA_T* b, ORA_T* out, uint64_t n, ORA_T q,
                     ORA_T bit, ORA_T mu)
{
    for (uint64_t i = 0; i < n; i++)
        out[i] = ORA_(mult)(a[i], b[i], q, bit, mu);
}

/* ---- schoolbook_poly_multiplication  ntt_cpu.cu:10-52 */
int ORA_(schoolbook)(const ORA_T* a, const ORA_T* b, ORA_T* out, int length, int poly,
                     ORA_T q, ORA_T bit, ORA_T mu)
{
    ORA_T* mv = (ORA_T*) calloc((size_t) length * 2, sizeof(ORA_T));
    if (!mv)
        return -1;
    for (int i = 0; i < length; i++)
        for (int j = 0; j < length; j++)
        {
            ORA_T m = ORA_(mult)(a[i], b[j], q, bit, mu);
            mv[i + j] = ORA_(add)(mv[i + j], m, q);
        }
    if (poly == 1)
        for (int i = 0; i < length; i++)
            out[i] = ORA_(add)(mv[i], mv[i + length], q);
    else
        for (int i = 0; i < length; i++)
            out[i] = ORA_(sub)(mv[i], mv[i + length], q);
    free(mv);
    return 0;
}

/* ---- NTTParameters4Step<T> pools  nttparameters.cu:229-303 ; shapes :305-354
 * out = {q, omega, psi}, n1, n2.  returns -1 for logn outside 12..24. */
int ORA_(fourstep_pool)(int logn, ORA_T* q_out, ORA_T* omega_out, ORA_T* psi_out,
                        int* n1, int* n2)
{
#if ORA_BITS == 32
    static const ORA_T primes[] = {268460033, 268582913, 268664833, 268369921,
                                   269221889, 269221889, 270532609, 270532609,
                                   270532609, 377487361, 377487361, 469762049,
                                   469762049};
    static const ORA_T W[] = {36747374, 249229369, 4092529, 175218169, 10653696,
                              238764304, 240100,   23104,   179776,    19321,
                              38809,    1600,      169};
    static const ORA_T PSI[] = {77090, 15787, 2023, 13237, 3264, 15452, 490,
                                152,   424,   139,  197,   40,   13};
#else
    static const ORA_T primes[] = {
        576460752303415297ULL, 576460752303439873ULL, 576460752304439297ULL,
        576460752308273153ULL, 576460752308273153ULL, 576460752315482113ULL,
        576460752315482113ULL, 576460752340123649ULL, 576460752364240897ULL,
        576460752475389953ULL, 576460752597024769ULL, 576460753024843777ULL,
        576460753175838721ULL};
    static const ORA_T W[] = {
        288482366111684746ULL, 37048445140799662ULL,  459782973201979845ULL,
        64800917766465203ULL,  425015386842055933ULL, 18734847765732801ULL,
        119109113519742895ULL, 227584740857897520ULL, 477282059544659462ULL,
        570131728462077067ULL, 433594414095420776ULL, 219263994987749328ULL,
        189790554094222112ULL};
    static const ORA_T PSI[] = {
        238394956950829ULL, 54612008597396ULL, 8242615629351ULL, 16141297350887ULL,
        3760097055997ULL,   11571974431275ULL, 328867687796ULL,  2298846063117ULL,
        731868219707ULL,    409596963254ULL,   189266227206ULL,  31864818375ULL,
        92067739764ULL};
#endif
    static const int shape[13][2] = {{32, 128},    {32, 256},    {32, 512},   {64, 512},
                                     {128, 512},   {32, 4096},   {32, 8192},  {32, 16384},
                                     {32, 32768},  {64, 32768},  {128, 32768}, {128, 65536},
                                     {256, 65536}};
    if (logn < 12 || logn > 24)
        return -1;
    *q_out = primes[logn - 12];
    *omega_out = W[logn - 12];
    *psi_out = PSI[logn - 12];
    *n1 = shape[logn - 12][0];
    *n2 = shape[logn - 12][1];
    return 0;
}

/* ---- small_forward/inverse_root_of_unity_table_generator  nttparameters.cu:356-380,398-428
 * natural-order tables of n1/2 and n2/2 powers of root^(n/n1) and root^(n/n2)
 * (inverse: of the modular inverse of those). */
void ORA_(fourstep_small_tables)(ORA_T root, uint64_t n, int n1, int n2, int inverse,
                                 ORA_T q, ORA_T bit, ORA_T mu, ORA_T* n1_table,
                                 ORA_T* n2_table)
{
    ORA_T exp_n1 = (ORA_T) (int) (n / (uint64_t) n1);
    ORA_T r1 = ORA_(exp)(root, exp_n1, q, bit, mu);
    if (inverse)
        r1 = ORA_(modinv)(r1, q, bit, mu);
    n1_table[0] = 1;
    for (int i = 1; i < (n1 >> 1); i++)
        n1_table[i] = ORA_(mult)(n1_table[i - 1], r1, q, bit, mu);

    ORA_T exp_n2 = (ORA_T) (int) (n / (uint64_t) n2);
    ORA_T r2 = ORA_(exp)(root, exp_n2, q, bit, mu);
    if (inverse)
        r2 = ORA_(modinv)(r2, q, bit, mu);
    n2_table[0] = 1;
    for (int i = 1; i < (n2 >> 1); i++)
        n2_table[i] = ORA_(mult)(n2_table[i - 1], r2, q, bit, mu);
}

/* ---- TW_forward_table_generator  nttparameters.cu:382-396 :
 *        W[i*n2+j] = root^(bitreverse(i, log2 n1) * j)
 *      TW_inverse_table_generator  nttparameters.cu:430-444 :
 *        W[i*n2+j] = inv_root^(bitreverse(j, log2 n2) * i)
 * `root` is root_of_unity (forward) or inverse_root_of_unity (inverse). */
void ORA_(fourstep_W)(ORA_T root, int n1, int n2, int inverse, ORA_T q, ORA_T bit,
                      ORA_T mu, ORA_T* W)
{
    int lg = inverse ? (int) log2((double) n2) : (int) log2((double) n1);
    for (int i = 0; i < n1; i++)
        for (int j = 0; j < n2; j++)
        {
            ORA_T index;
            if (!inverse)
            {
                index = (ORA_T) ora_bitreverse(i, lg);
                index = index * (ORA_T) j;
            }
            else
            {
                index = (ORA_T) ora_bitreverse(j, lg);
                index = index * (ORA_T) i;
            }
            W[(size_t) i * n2 + j] = ORA_(exp)(root, index, q, bit, mu);
        }
}

/* ---- NTT_4STEP_CPU<T>::core_ntt  src/lib/ntt_4step/ntt_4step_cpu.cu:116-154
 * cyclic CT with natural-order table indexed bitreverse(i, log_size-1). */
static void ORA_(core_ntt)(ORA_T* input, const ORA_T* root_table, int log_size,
                           ORA_T q, ORA_T bit, ORA_T mu)
{
    int n_ = 1 << log_size;
    int t = n_;
    int m = 1;
    while (m < n_)
    {
        t = t >> 1;
        for (int i = 0; i < m; i++)
        {
            int j1 = 2 * i * t;
            int j2 = j1 + t - 1;
            int index = ora_bitreverse(i, log_size - 1);
            ORA_T S = root_table[index];
            for (int j = j1; j < (j2 + 1); j++)
            {
                ORA_T U = input[j];
                ORA_T V = ORA_(mult)(input[j + t], S, q, bit, mu);
                input[j] = ORA_(add)(U, V, q);
                input[j + t] = ORA_(sub)(U, V, q);
            }
        }
        m = m << 1;
    }
}

/* ---- NTT_4STEP_CPU<T>::core_intt  ntt_4step_cpu.cu:155-197 */
static void ORA_(core_intt)(ORA_T* input, const ORA_T* root_table, int log_size,
                            ORA_T q, ORA_T bit, ORA_T mu)
{
    int n_ = 1 << log_size;
    int t = 1;
    int m = n_;
    while (m > 1)
    {
        int j1 = 0;
        int h = m >> 1;
        for (int i = 0; i < h; i++)
        {
            int j2 = j1 + t - 1;
            int index = ora_bitreverse(i, log_size - 1);
            ORA_T S = root_table[index];
            for (int j = j1; j < (j2 + 1); j++)
            {
                ORA_T U = input[j];
                ORA_T V = input[j + t];
                input[j] = ORA_(add)(U, V, q);
                input[j + t] = ORA_(sub)(U, V, q);
                input[j + t] = ORA_(mult)(input[j + t], S, q, bit, mu);
            }
            j1 = j1 + (t << 1);
        }
        t = t << 1;
        m = m >> 1;
    }
}

/* rows x cols (row-major) -> cols x rows ; restates transpose_matrix ntt_4step_cpu.cu:266-285
 * on flat storage (the reference's vector<vector<T>> holds the same values). */
static void ORA_(transpose)(const ORA_T* in, ORA_T* out, int rows, int cols)
{
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            out[(size_t) j * rows + i] = in[(size_t) i * cols + j];
}

/* ---- NTT_4STEP_CPU<T>::intt_first_transpose / vector_to_matrix_intt
 * ntt_4step_cpu.cu:229-245, 287-299 :  flat[i*cols + j] = array[i + j*rows], rows=n1, cols=n2 */
void ORA_(fourstep_intt_first_transpose)(const ORA_T* in, ORA_T* out, int n1, int n2)
{
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j)
            out[(size_t) i * n2 + j] = in[(size_t) i + (size_t) j * n1];
}

/* ---- NTT_4STEP_CPU<T>::ntt  ntt_4step_cpu.cu:33-68
 * in (n1 x n2) -> transpose (n2 x n1) -> n1-point NTT on each of the n2 rows ->
 * transpose back (n1 x n2) -> * W -> n2-point NTT on each of the n1 rows ->
 * transpose (n2 x n1) -> flat.  Tables are NATURAL order (n1/2, n2/2 entries), W has n entries. */
int ORA_(fourstep_ntt)(const ORA_T* in, ORA_T* out, int n1, int n2, const ORA_T* n1_table,
                       const ORA_T* n2_table, const ORA_T* W, ORA_T q, ORA_T bit,
                       ORA_T mu)
{
    size_t n = (size_t) n1 * n2;
    int lg1 = (int) log2((double) n1), lg2 = (int) log2((double) n2);
    ORA_T* a = (ORA_T*) malloc(n * sizeof(ORA_T));
    ORA_T* b = (ORA_T*) malloc(n * sizeof(ORA_T));
    if (!a || !b)
    {
        free(a);
        free(b);
        return -1;
    }
    ORA_(transpose)(in, a, n1, n2); /* a: n2 x n1 */
    for (int i = 0; i < n2; i++)
        ORA_(core_ntt)(a + (size_t) i * n1, n1_table, lg1, q, bit, mu);
    ORA_(transpose)(a, b, n2, n1); /* b: n1 x n2 */
    for (size_t i = 0; i < n; i++) /* product(): ntt_4step_cpu.cu:201-211 */
        b[i] = ORA_(mult)(b[i], W[i], q, bit, mu);
    for (int i = 0; i < n1; i++)
        ORA_(core_ntt)(b + (size_t) i * n2, n2_table, lg2, q, bit, mu);
    ORA_(transpose)(b, out, n1, n2); /* out: n2 x n1 */
    free(a);
    free(b);
    return 0;
}

/* ---- NTT_4STEP_CPU<T>::intt  ntt_4step_cpu.cu:70-111 */
int ORA_(fourstep_intt)(const ORA_T* in, ORA_T* out, int n1, int n2,
                        const ORA_T* n1_inv_table, const ORA_T* n2_inv_table,
                        const ORA_T* W_inv, ORA_T n_inv, ORA_T q, ORA_T bit, ORA_T mu)
{
    size_t n = (size_t) n1 * n2;
    int lg1 = (int) log2((double) n1), lg2 = (int) log2((double) n2);
    ORA_T* a = (ORA_T*) malloc(n * sizeof(ORA_T));
    ORA_T* b = (ORA_T*) malloc(n * sizeof(ORA_T));
    if (!a || !b)
    {
        free(a);
        free(b);
        return -1;
    }
    ORA_(fourstep_intt_first_transpose)(in, a, n1, n2); /* a: n2 rows of n1 */
    for (int i = 0; i < n2; i++)
        ORA_(core_intt)(a + (size_t) i * n1, n1_inv_table, lg1, q, bit, mu);
    ORA_(transpose)(a, b, n2, n1); /* b: n1 x n2 */
    for (size_t i = 0; i < n; i++)
        b[i] = ORA_(mult)(b[i], W_inv[i], q, bit, mu);
    for (int i = 0; i < n1; i++)
        ORA_(core_intt)(b + (size_t) i * n2, n2_inv_table, lg2, q, bit, mu);
    ORA_(transpose)(b, out, n1, n2);
    for (size_t i = 0; i < n; i++)
        out[i] = ORA_(mult)(out[i], n_inv, q, bit, mu);
    free(a);
    free(b);
    return 0;
}

/* ---- portable synthetic input (NOT from the reference):
 * x[k] = splitmix64(seed ^ (offset + k)) mod q */
void ORA_(splitmix_fill)(uint64_t seed, uint64_t offset, uint64_t count, ORA_T q,
                         ORA_T* out)
{
    for (uint64_t k = 0; k < count; k++)
        out[k] = (ORA_T) (ora_splitmix64(seed ^ (offset + k)) % (uint64_t) q);
}

/* ---- batch driver for the cpu_baseline timing leg: `batch` independent forward
 * (dir=0) or inverse (dir=1) Merge transforms, poly p uses modulus/table slot p % mod_count;
 * tables are NATURAL order, slot stride = table_stride elements. Threads via OpenMP. */
void ORA_(merge_batch)(ORA_T* data, int batch, int logn, int poly, int dir,
                       const ORA_T* tables, uint64_t table_stride, const ORA_T* q,
                       int mod_count, int nthreads)
{
    size_t n = (size_t) 1 << logn;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int p = 0; p < batch; p++)
    {
        int mi = p % mod_count;
        ORA_T bit, mu;
        ORA_(modulus)(q[mi], &bit, &mu);
        if (dir == 0)
            ORA_(merge_ntt)(data + p * n, logn, poly, tables + mi * table_stride, q[mi],
                            bit, mu);
        else
            ORA_(merge_intt)(data + p * n, logn, poly, tables + mi * table_stride, q[mi],
                             bit, mu);
    }
    (void) nthreads;
}
