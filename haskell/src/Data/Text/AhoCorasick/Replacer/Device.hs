{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE MagicHash                #-}
-- | Device back end of "Data.Text.AhoCorasick.Replacer" (reference: src/Data/Text/AhoCorasick/Replacer.hs):
-- 'run' (:200-201) and 'runWithLimit' (:203-242) for batches of haystacks.  The whole loop of @runWithLimit.go@ --
-- one scan per priority that matches, @prependMatch@ / @makeMatch@ (:252-274), the @replacementLength@ check
-- (:183-187, :240), @sort@ + @removeOverlap@ (:191-198) and @replace@ (:163-180) -- runs in HBM; only the finished
-- texts come back.  'Replacer.build', 'compose', 'mapReplacement', 'setCaseSensitivity' stay the reference's: a
-- 'DeviceReplacer' is made from whatever 'Replacer.Replacer' they produced.
module Data.Text.AhoCorasick.Replacer.Device
  ( DeviceReplacer
  , toDevice
  , run
  , runWithLimit
  ) where

import Data.Int (Int64)
import Data.Word (Word32, Word64, Word8)
import Foreign
import Foreign.C.Types

import qualified Data.Primitive.ByteArray as BA
import qualified Data.Text.Array as TextArray
import qualified Data.IntMap.Strict as IntMap
import qualified Data.Vector as Vector

import Data.Text.AhoCorasick.Automaton (AcMachine (..), CodeUnitIndex (..))
import Data.Text.Utf8 (Text (..))

import qualified Data.Text.AhoCorasick.Automaton.Device as Dev
import qualified Data.Text.AhoCorasick.Replacer as Replacer
import qualified Data.Text.AhoCorasick.Searcher as Searcher
import qualified Data.Text.Utf8 as Utf8

data AmReplacer
data AmReplaced

-- | @am_payload {i64 priority; u32 len_bytes; u32 len_code_points; u64 repl_off; u32 repl_len; u32 reserved}@ = 'Replacer.Payload' (Replacer.hs:59-70)
data AmPayload = AmPayload !Int64 !Word32 !Word32 !Word64 !Word32

instance Storable AmPayload where
  sizeOf _ = 32
  alignment _ = 8
  poke p (AmPayload pr lb lc off len) =
    pokeByteOff p 0 pr >> pokeByteOff p 8 lb >> pokeByteOff p 12 lc >> pokeByteOff p 16 off >> pokeByteOff p 24 len >> pokeByteOff p 28 (0 :: Word32)
  peek p = AmPayload <$> peekByteOff p 0 <*> peekByteOff p 8 <*> peekByteOff p 12 <*> peekByteOff p 16 <*> peekByteOff p 24

-- safe: the call flattens nothing itself but uploads tables and may wait for the IgnoreCase image (milliseconds to a second): an unsafe call would hold the capability and stall GC
foreign import ccall safe "am_replacer_create"
  c_am_replacer_create :: Ptr Dev.AmAutomaton -> CInt -> Ptr Word64 -> Ptr Word32 -> Ptr AmPayload -> CSize
                       -> Ptr Word8 -> CSize -> Int64 -> Ptr (Ptr AmReplacer) -> IO CInt
foreign import ccall unsafe "&am_replacer_destroy"
  p_am_replacer_destroy :: FunPtr (Ptr AmReplacer -> IO ())
foreign import ccall safe "am_replacer_run"
  c_am_replacer_run :: Ptr AmReplacer -> Ptr Dev.AmSlice -> CSize -> Word64 -> Ptr (Ptr AmReplaced) -> IO CInt
foreign import ccall unsafe "am_replaced_get"
  c_am_replaced_get :: Ptr AmReplaced -> CSize -> Ptr (Ptr Word8) -> Ptr CSize -> IO CInt   -- 1 = Just, 0 = Nothing
foreign import ccall unsafe "am_replaced_free"
  c_am_replaced_free :: Ptr AmReplaced -> IO ()

data DeviceReplacer = DeviceReplacer
  { drMachine :: !(Dev.DeviceMachine Replacer.Payload)      -- keeps the automaton alive: the replacer handle borrows it
  , drHandle  :: !(ForeignPtr AmReplacer)
  }

-- | After 'Replacer.build' / 'compose' / 'setCaseSensitivity' (unchanged): the automaton's arrays, 'machineValues' in
-- flat form, the payloads with their replacements in one blob, and @minPriority = 1 - numNeedles@ (Replacer.hs:217).
-- Payload k of the table is the k-th (needle, payload) pair of the searcher; a state's value list names them by index.
toDevice :: Replacer.Replacer -> IO DeviceReplacer
toDevice (Replacer.Replacer s) = do
  dm <- Dev.toDevice (Searcher.automaton s)
  let values   = machineValues (Dev.dmMachine dm)
      payloads = map snd (Searcher.needles s)
      -- priorities are distinct (build: 0, -1, -2, ...; compose keeps them distinct): a payload's index is looked up by its priority in a map built once
      -- (a linear search per value made a 100k-needle replacer quadratic: ADVICE r5)
      prioIndex = IntMap.fromList (zip (map (fromIntegral . Replacer.needlePriority) payloads) [0 :: Int ..])
      indexOf p = IntMap.findWithDefault (error "Replacer.toDevice: a value's payload is not among the needles") (fromIntegral (Replacer.needlePriority p)) prioIndex
      offsets  = scanl (+) 0 (map (fromIntegral . length) (Vector.toList values)) :: [Word64]
      flat     = [ fromIntegral (indexOf p) | ps <- Vector.toList values, p <- ps ] :: [Word32]
      repls    = map (Utf8.unpackUtf8 . Replacer.needleReplacement) payloads
      replOffs = scanl (+) 0 (map (fromIntegral . length) repls) :: [Word64]
      table    = [ AmPayload (fromIntegral (Replacer.needlePriority p)) (fromIntegral (codeUnitIndex (Replacer.needleLengthBytes p)))
                             (fromIntegral (Replacer.needleLengthCodePoints p)) off (fromIntegral (length r))
                 | (p, off, r) <- zip3 payloads replOffs repls ]
      blob     = concat repls
      minPrio  = 1 - fromIntegral (Searcher.numNeedles s) :: Int64
  withArray offsets $ \pOff -> withArray (if null flat then [0] else flat) $ \pVals ->
    withArrayLen table $ \nP pP -> withArrayLen (if null blob then [0] else blob) $ \_ pBlob ->
    withForeignPtr (Dev.dmHandle dm) $ \ph -> alloca $ \out -> do
      c_am_replacer_create ph (Dev.caseFlag (Replacer.replacerCaseSensitivity (Replacer.Replacer s))) pOff pVals pP (fromIntegral nP)
                           pBlob (fromIntegral (length blob)) minPrio out >>= Dev.checkRc
      DeviceReplacer dm <$> (peek out >>= newForeignPtr p_am_replacer_destroy)

-- | 'Replacer.run' (Replacer.hs:200-201) for a batch: @runWithLimit maxBound@.
run :: DeviceReplacer -> [Text] -> IO [Text]
run r texts = map (maybe (error "Replacer.run: no limit, yet a result is missing") id) <$> runWithLimit r maxBound texts

-- | 'Replacer.runWithLimit' (Replacer.hs:203-242) for a batch: 'Nothing' where the text would grow beyond @maxLength@ code units.
runWithLimit :: DeviceReplacer -> CodeUnitIndex -> [Text] -> IO [Maybe Text]
runWithLimit (DeviceReplacer dm h) (CodeUnitIndex maxLength) texts =
  Dev.withPinnedTexts texts $ \pSlices n ->
  withForeignPtr (Dev.dmHandle dm) $ \_ -> withForeignPtr h $ \ph -> alloca $ \out -> do
    let limit = if maxLength == maxBound then maxBound else fromIntegral (max 0 maxLength) :: Word64
    c_am_replacer_run ph pSlices (fromIntegral n) limit out >>= Dev.checkRc
    res <- peek out
    texts' <- mapM (fetch res) [0 .. n - 1]
    c_am_replaced_free res
    pure texts'
  where
    fetch res i = alloca $ \pp -> alloca $ \pl -> do
      just <- c_am_replaced_get res (fromIntegral i) pp pl
      Dev.checkRc (min just 0)
      if just == 0 then pure Nothing else do
        p <- peek pp
        len <- fromIntegral <$> peek pl
        mb <- BA.newByteArray len
        BA.copyPtrToMutableByteArray mb 0 p len
        BA.ByteArray ba# <- BA.unsafeFreezeByteArray mb
        pure (Just (Text (TextArray.ByteArray ba#) 0 len))      -- (as Utf8.fromByteList builds a Text, Utf8.hs:164-166)
