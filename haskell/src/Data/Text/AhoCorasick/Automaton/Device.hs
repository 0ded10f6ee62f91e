{-# LANGUAGE BangPatterns             #-}
{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE MagicHash                #-}
{-# LANGUAGE ScopedTypeVariables      #-}
-- | Device back end of "Data.Text.AhoCorasick.Automaton" (reference: src/Data/Text/AhoCorasick/Automaton.hs).
--
-- What stays in Haskell: 'Aho.build' (Automaton.hs:176-200: state numbering, fallback edges, 'machineValues' and
-- their order), the fold function with 'Step' / 'Done', the payload values @v@, and 'Data.Char.toLower'.
-- What moves to libam: the body of 'Aho.runWithCase' (Automaton.hs:442-534) -- decode, lower-case, transition
-- lookup, "is anything reported here" -- for whole batches of haystacks on an MI355X.
--
-- libam answers with one record per (haystack, end position) at which the reference calls the fold function at
-- least once: @AmMatch endPos haystack state@, where @machineValues ! state@ is exactly the list the reference
-- folds there, in its order (include/am.h, "Result format").  Records are sorted by (haystack, endPos) = the
-- order of the reference's left fold, so folding them reproduces 'runWithCase' including its early exit.
--
-- The FFI shape follows the reference's only precedent, benchmark/rust-ffi/app/Main.hs:28-52: slices as
-- {ptr, off, len}, pinned byte arrays, the callee borrows.
module Data.Text.AhoCorasick.Automaton.Device
  ( DeviceMachine (..)
  , toDevice
  , runWithCaseDevice
  , runTextDevice
  , runLowerDevice
  , runBatchWithCaseDevice
  , countMatchesDevice
  , containsAnyDevice
    -- * For the sibling modules
  , AmAutomaton
  , AmSlice (..)
  , AmMatch (..)
  , caseFlag
  , checkRc
  , lowerPairs
  , withPrimArray
  , withPinnedText
  , withPinnedTexts
  ) where

import Control.Exception (ErrorCall (..), throwIO)
import Control.Monad (when)
import Data.Char (ord)
import Data.Word (Word32, Word64, Word8)
import Foreign
import Foreign.C.String (peekCString)
import Foreign.C.Types

import qualified Data.Char as Char
import qualified Data.Primitive as Prim
import qualified Data.Primitive.ByteArray as BA
import qualified Data.Text.Array as TextArray
import qualified Data.Vector as Vector

import Data.Text.AhoCorasick.Automaton (AcMachine (..), CaseSensitivity (..), CodeUnitIndex (..), Match (..), Next (..))
import Data.Text.Utf8 (Text (..))

data AmAutomaton
data AmMatches

-- | @am_slice@ = @U8Slice@ of benchmark/rust-ffi/app/Main.hs:34-45 (= Data.Text.Internal.Text array/offset/length).
data AmSlice = AmSlice !(Ptr Word8) !CSize !CSize

instance Storable AmSlice where
  sizeOf _ = 24
  alignment _ = 8
  poke p (AmSlice a o l) = pokeByteOff p 0 a >> pokeByteOff p 8 o >> pokeByteOff p 16 l
  peek p = AmSlice <$> peekByteOff p 0 <*> peekByteOff p 8 <*> peekByteOff p 16

-- | @am_match {u64 end_pos; u32 haystack; u32 state}@
data AmMatch = AmMatch !Word64 !Word32 !Word32

instance Storable AmMatch where
  sizeOf _ = 16
  alignment _ = 8
  peek p = AmMatch <$> peekByteOff p 0 <*> peekByteOff p 8 <*> peekByteOff p 12
  poke p (AmMatch e h s) = pokeByteOff p 0 e >> pokeByteOff p 8 h >> pokeByteOff p 12 s

-- am_automaton_create_ex = am_automaton_create + the host's own lower-casing as data (see 'lowerPairs')
-- safe: the call flattens the automaton (hundreds of milliseconds for 100k needles, a DFA table of up to 1 GiB for a dictionary): unsafe would block the capability and GC
foreign import ccall safe "am_automaton_create_ex"
  c_am_automaton_create_ex :: Ptr Word64 -> CSize -> Ptr Word32 -> CSize -> Ptr Word64 -> Ptr Word32
                           -> Ptr Word32 -> Ptr Word32 -> CSize -> Ptr (Ptr AmAutomaton) -> IO CInt
foreign import ccall unsafe "&am_automaton_destroy"
  p_am_automaton_destroy :: FunPtr (Ptr AmAutomaton -> IO ())
-- the run entry points block for the duration of the GPU call: `safe`, inputs must be pinned
foreign import ccall safe "am_run"
  c_am_run :: Ptr AmAutomaton -> CInt -> Ptr AmSlice -> CSize -> Ptr (Ptr AmMatches) -> IO CInt
foreign import ccall safe "am_count"
  c_am_count :: Ptr AmAutomaton -> CInt -> Ptr AmSlice -> CSize -> Ptr Word64 -> IO CInt
foreign import ccall safe "am_contains_any"
  c_am_contains_any :: Ptr AmAutomaton -> CInt -> Ptr AmSlice -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall unsafe "am_matches_size"
  c_am_matches_size :: Ptr AmMatches -> IO Word64
foreign import ccall safe "am_matches_data"
  c_am_matches_data :: Ptr AmMatches -> IO (Ptr AmMatch)
foreign import ccall unsafe "am_matches_free"
  c_am_matches_free :: Ptr AmMatches -> IO ()
foreign import ccall unsafe "am_last_error"
  c_am_last_error :: IO (Ptr CChar)

-- | An 'AcMachine' plus its flattened copy in HBM.  'machineValues' never leaves Haskell.
data DeviceMachine v = DeviceMachine
  { dmMachine :: !(AcMachine v)
  , dmHandle  :: !(ForeignPtr AmAutomaton)
  }

-- | What THIS build's 'Data.Char.toLower' does, as data: IgnoreCase lower-cases the haystack with
-- @Utf8.lowerCodePoint@ (Utf8.hs:145-151 = 'Char.toLower' above ASCII), and base's table follows the GHC version
-- (Unicode 15 for 9.6-9.10, 16 from 9.12).  Computed once per process (a CAF; 1.1 M 'toLower' calls, ~1 400-1 500
-- pairs); libam bakes it into the IgnoreCase image, so the device lowers exactly like the host whatever compiler
-- built it.  (NULL / 0 instead = libam's built-in Unicode 14.0 table.)
lowerPairs :: ([Word32], [Word32])
lowerPairs = unzip
  [ (fromIntegral (ord c), fromIntegral (ord l))
  | c <- [minBound .. maxBound], let l = Char.toLower c, l /= c ]
{-# NOINLINE lowerPairs #-}

-- | Second half of 'Aho.build' (Automaton.hs:176-200): hand the packed arrays to libam.
toDevice :: AcMachine v -> IO (DeviceMachine v)
toDevice m@(AcMachine values transitions offsets rootAscii) =
  withPrimArray transitions $ \pT nT ->
  withPrimArray offsets     $ \pO nO ->
  withPrimArray rootAscii   $ \pR _  ->
  withArrayLen (map (fromIntegral . length) (Vector.toList values)) $ \_ pV ->
  withArrayLen (fst lowerPairs) $ \nL pLF -> withArray (snd lowerPairs) $ \pLT ->
  alloca $ \out -> do
    rc <- c_am_automaton_create_ex (castPtr pT) (fromIntegral nT) (castPtr pO) (fromIntegral (nO - 1)) (castPtr pR) pV
                                   pLF pLT (fromIntegral nL) out
    checkRc rc
    DeviceMachine m <$> (peek out >>= newForeignPtr p_am_automaton_destroy)

caseFlag :: CaseSensitivity -> CInt
caseFlag CaseSensitive = 0   -- AM_CASE_SENSITIVE
caseFlag IgnoreCase    = 1   -- AM_IGNORE_CASE

-- | Drop-in for 'Aho.runWithCase' (Automaton.hs:443): same fold, same order, same early exit.
runWithCaseDevice :: CaseSensitivity -> a -> (a -> Match v -> Next a) -> DeviceMachine v -> Text -> IO a
runWithCaseDevice cs seed f dm text = head <$> runBatchWithCaseDevice cs seed f dm [text]

-- | 'Aho.runText' (Automaton.hs:539-541) / 'Aho.runLower' (:551-553).
runTextDevice, runLowerDevice :: a -> (a -> Match v -> Next a) -> DeviceMachine v -> Text -> IO a
runTextDevice  = runWithCaseDevice CaseSensitive
runLowerDevice = runWithCaseDevice IgnoreCase

-- | The same for a batch: ONE device call; the records come back sorted by (haystack, end position) and every
-- haystack's run is folded like 'collectMatches' (Automaton.hs:522-534) folds it, stopping at that haystack's 'Done'.
runBatchWithCaseDevice :: CaseSensitivity -> a -> (a -> Match v -> Next a) -> DeviceMachine v -> [Text] -> IO [a]
runBatchWithCaseDevice cs seed f (DeviceMachine m h) texts =
  withPinnedTexts texts $ \pSlices nTexts ->
  alloca $ \out -> withForeignPtr h $ \ph -> do
    c_am_run ph (caseFlag cs) pSlices (fromIntegral nTexts) out >>= checkRc
    ms <- peek out
    n  <- fromIntegral <$> c_am_matches_size ms
    p  <- c_am_matches_data ms
    when (n > 0 && p == nullPtr) (c_am_matches_free ms >> checkRc (-3))
    let -- fold the records [i, ..) that belong to haystack `hay`; returns the accumulator and the first record behind them
        goHay :: Word32 -> Int -> a -> Bool -> IO (a, Int)
        goHay !hay !i !acc !done
          | i >= n = pure (acc, i)
          | otherwise = do
              AmMatch pos hay' st <- peekElemOff p i
              if hay' /= hay then pure (acc, i)
              else if done then goHay hay (i + 1) acc True
              else case goVals (fromIntegral pos) (machineValues m Vector.! fromIntegral st) acc of
                     Step acc' -> goHay hay (i + 1) acc' False
                     Done acc' -> goHay hay (i + 1) acc' True
        goVals !_ [] !acc = Step acc
        goVals !pos (v : vs) !acc = case f acc (Match (CodeUnitIndex pos) v) of
          Step acc' -> goVals pos vs acc'
          Done acc' -> Done acc'
        goAll !hay !i
          | hay >= fromIntegral nTexts = pure []
          | otherwise = do
              (acc, i') <- goHay hay i seed False
              (acc :) <$> goAll (hay + 1) i'
    r <- goAll 0 0
    c_am_matches_free ms
    pure r

-- | benchmark/haskell/app/Main.hs:67-76 @countMatches@, for a batch: one launch, the counts come back per haystack.
countMatchesDevice :: CaseSensitivity -> DeviceMachine v -> [Text] -> IO [Word64]
countMatchesDevice cs (DeviceMachine _ h) texts =
  withPinnedTexts texts $ \pSlices n ->
  allocaArray (max n 1) $ \pCounts -> withForeignPtr h $ \ph -> do
    c_am_count ph (caseFlag cs) pSlices (fromIntegral n) pCounts >>= checkRc
    peekArray n pCounts

-- | 'Data.Text.AhoCorasick.Searcher.containsAny' (Searcher.hs:156-164), for a batch (the scan of a haystack stops at its first match).
containsAnyDevice :: CaseSensitivity -> DeviceMachine v -> [Text] -> IO [Bool]
containsAnyDevice cs (DeviceMachine _ h) texts =
  withPinnedTexts texts $ \pSlices n ->
  allocaArray (max n 1) $ \pFlags -> withForeignPtr h $ \ph -> do
    c_am_contains_any ph (caseFlag cs) pSlices (fromIntegral n) pFlags >>= checkRc
    map (/= (0 :: Word8)) <$> peekArray n pFlags

-- ---- helpers -----------------------------------------------------------------------------------

-- | Negative return code -> exception carrying @am_last_error()@ (thread-local: Haskell threads that call the `safe`
-- imports should be bound -- 'Control.Concurrent.forkOS' -- so that the message, and libam's per-thread stream, are theirs).
checkRc :: CInt -> IO ()
checkRc rc = when (rc < 0) $ do
  msg <- c_am_last_error >>= peekCString
  throwIO (ErrorCall ("libam: " ++ msg ++ " (code " ++ show rc ++ ")"))

-- | The payload of a 'Prim.PrimArray' as a pointer that stays valid during a `safe` call: pinned arrays are used in
-- place (Utf8.hs:326-327 isArrayPinned), others are copied into a pinned one first (the GC may move unpinned arrays).
withPrimArray :: forall a b. Prim.Prim a => Prim.PrimArray a -> (Ptr a -> Int -> IO b) -> IO b
withPrimArray arr@(Prim.PrimArray ba#) k = do
  let n = Prim.sizeofPrimArray arr
      bytes = n * Prim.sizeOf (undefined :: a)
      src = BA.ByteArray ba#
  pinned <- if BA.isByteArrayPinned src then pure src else do
    mb <- BA.newPinnedByteArray bytes
    BA.copyByteArray mb 0 src 0 bytes
    BA.unsafeFreezeByteArray mb
  r <- k (castPtr (BA.byteArrayContents pinned)) n
  Prim.touch pinned            -- keeps the array alive until the call has returned
  pure r

-- | One haystack as an 'AmSlice' {ptr, off, len} (Text = array + offset + length in code units, Utf8.hs:96-105).
withPinnedText :: Text -> (AmSlice -> IO b) -> IO b
withPinnedText (Text (TextArray.ByteArray ba#) off len) k =      -- (the constructor the reference itself takes apart, Utf8.hs:326-331)
  withPrimArray (Prim.PrimArray ba# :: Prim.PrimArray Word8) $ \p _ -> k (AmSlice p (fromIntegral off) (fromIntegral len))

-- | A batch: one 'AmSlice' per text, all alive for the duration of the call.
withPinnedTexts :: [Text] -> (Ptr AmSlice -> Int -> IO b) -> IO b
withPinnedTexts texts k = go texts []
  where
    go [] acc = withArrayLen (reverse acc) (\n p -> k p n)
    go (t : ts) acc = withPinnedText t (\s -> go ts (s : acc))
